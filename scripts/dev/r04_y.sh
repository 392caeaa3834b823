#!/bin/bash
# round 4, call y: second batch of instruction trims in the team kernel (transposed DPP stages of the wavefront reduction,
# C = (1 - A) / x, quarter-angle series, normal equations in units of f, the square roots of norm_greater behind a branch):
# parity tests, sweep, then same-box A/B against the previous commit and the one before the first batch
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04y
timeout 900 python -m pytest tests/test_gpu_semantics.py tests/test_gpu_parity.py tests/test_gpu_backward.py tests/test_gpu_edge.py tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python scripts/dev/sweep.py 600 2>&1 | tail -2
run() { timeout 300 python bench.py --steps 600 --warmup 60 --no-cpu-baseline --no-training --batch 0 --no-exact | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('value %.0f ms %.4f refine stage %.4f' % (d['value'], d['ms_per_step'], d['phase_ms']['refine']))"; }
for rep in 1 2 3; do
echo "== now"; unset ESAC_HIP_LIB; run
echo "== first batch (28eadbe)"; export ESAC_HIP_LIB=$GRAFT_REPO_ROOT/scratch/lib_trim1.so; run
echo "== before (23a5945)"; export ESAC_HIP_LIB=$GRAFT_REPO_ROOT/scratch/lib_head.so; run
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04y/ab.txt
