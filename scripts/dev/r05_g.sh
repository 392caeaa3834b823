#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r05/g_tests.txt
cat gpurun_out/r05/g_tests.txt
python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-training --batch 0 > gpurun_out/r05/g_bench.json 2> gpurun_out/r05/g_bench.err
tail -3 gpurun_out/r05/g_bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05/g_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'seed1305', d.get('value_seed1305'))
print('value_fast', d.get('value_fast'))
print('sharded', d.get('sharded_world1'))
print('phase', d.get('phase_ms'))
for k in d.get('kernels', []): print(k['stage'], k['avg_us'])
print('roofline', {k: d['roofline'][k] for k in ('kernel', 'achieved', 'frac', 'kernel_ms')})
print('with_h2d', d.get('with_h2d'))
PY
