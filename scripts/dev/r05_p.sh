#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_semantics.py tests/test_gpu_edge.py tests/test_gpu_parity.py tests/test_integration_doc.py tests/test_harness.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-training --batch 0 --no-exact | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value %.0f ms %.4f with_h2d %.0f ms %.4f' % (d['value'], d['ms_per_step'], d['with_h2d']['value'], d['with_h2d']['ms_per_call']))"; done
