#!/bin/bash
# round 5, call K: the refinement stage reads 66 or 70 us from process to process on the same box with the same library -- against what?
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for rep in 1 2 3 4 5 6 7 8; do
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-exact | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms %.4f' % d['ms_per_step'], {k['stage']: round(k['avg_us'],1) for k in d['kernels']}, d['refine']['xcd_census'], d['refine']['exchanges'])"
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/k_bimodal.txt
