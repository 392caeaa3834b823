#!/bin/bash
# round 5, call I: k_rescore with 16 wavefronts per hypothesis on the headline shape (A/B against the previous commit's library)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
run() { timeout 300 python bench.py --steps 600 --warmup 60 --no-cpu-baseline --no-extras --no-exact | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value %.0f ms %.4f seed1305 %.0f' % (d['value'], d['ms_per_step'], d['value_seed1305']), {k['stage']: round(k['avg_us'],1) for k in d['kernels']})"; }
for rep in 1 2 3; do
echo "== lib_head"; export ESAC_HIP_LIB=$GRAFT_REPO_ROOT/scratch/lib_head.so; run
echo "== tree"; unset ESAC_HIP_LIB; run
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/i_ab.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ref_golden.py tests/test_gpu_backward.py -m gpu -x -q 2>&1 | tail -4
