#!/bin/bash
# round 4: does the placement of the kernel-argument segment (HIP_FORCE_DEV_KERNARG) show in the latency shape?
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04k
run() { timeout 300 python bench.py --steps 600 --warmup 60 --no-cpu-baseline --no-training --batch 0 --no-exact | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('value %.0f ms %.4f phases %s' % (d['value'], d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items() if k in ('sample_p3p','score','select_rescore','refine')}))"; }
for rep in 1 2 3; do
for v in 0 1; do echo "== HIP_FORCE_DEV_KERNARG=$v"; HIP_FORCE_DEV_KERNARG=$v run; done
echo "== unset"; run
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04k/kernarg.txt
