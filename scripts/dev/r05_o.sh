#!/bin/bash
# round 5, call O: converged trials are not sent up the lambda ladder over a last-bit difference -- tests, sweep, first calls
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 600 python scripts/dev/sweep.py 800 2>&1 | grep -v amdgpu.ids | tail -4
python scripts/dev/call8_probe.py 2>&1 | grep -v amdgpu.ids | sed -n 6,14p
python scripts/dev/first_calls.py 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-extras --no-exact | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value %.0f ms %.4f seed1305 %.0f' % (d['value'], d['ms_per_step'], d['value_seed1305']), {k['stage']: round(k['avg_us'],1) for k in d['kernels']})"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-exact | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('driver-style value %.0f ms %.4f seed1305 %.0f' % (d['value'], d['ms_per_step'], d['value_seed1305']), {k['stage']: round(k['avg_us'],1) for k in d['kernels']})"
