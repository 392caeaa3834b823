#!/bin/bash
# round 5, call C: bounded / latched team time-out, slot-team route decided on the device, teams for small batches
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests/test_gpu_semantics.py tests/test_gpu_edge.py tests/test_gpu_backward.py tests/test_integration_doc.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r05/c_tests.txt
cat gpurun_out/r05/c_tests.txt
for B in 16 32 64; do
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-training --no-exact --batch $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('B=$B value %.0f ms %.4f batched %s' % (d['value'], d['ms_per_step'], d.get('batched')))"
ESAC_HIP_LIB=$GRAFT_REPO_ROOT/scratch/lib_r04.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-training --no-exact --batch $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('r04 B=$B value %.0f ms %.4f batched %s' % (d['value'], d['ms_per_step'], d.get('batched')))"
done 2>&1 | tee gpurun_out/r05/c_batch.txt
