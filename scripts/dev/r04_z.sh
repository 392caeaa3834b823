#!/bin/bash
# round 4, call z: LDS layout (arrays ahead of the pad / the list: no address materialised per access), normal equations
# without the structural-zero products: tests + same-box A/B, headline call and the batch of 16 (one workgroup per frame)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04z
timeout 900 python -m pytest tests/test_gpu_semantics.py tests/test_gpu_parity.py tests/test_gpu_backward.py tests/test_gpu_edge.py tests/test_gpu_ref_golden.py -x -q -m gpu 2>&1 | tail -3
run() { timeout 300 python bench.py --steps 600 --warmup 60 --no-cpu-baseline --no-training --batch 16 --no-exact | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('value %.0f ms %.4f refine stage %.4f batch16 %.0f' % (d['value'], d['ms_per_step'], d['phase_ms']['refine'], d['batched']['value']))"; }
for rep in 1 2 3; do
echo "== now"; unset ESAC_HIP_LIB; run
echo "== previous commit"; export ESAC_HIP_LIB=$GRAFT_REPO_ROOT/scratch/lib_trim3.so; run
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04z/ab.txt
