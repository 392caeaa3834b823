#!/bin/bash
# round 5, call H: A/B of (a) launches through hipFunction_t, (b) + the rotation series evaluated once in the lanes
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
run() { timeout 300 python bench.py --steps 600 --warmup 60 --no-cpu-baseline --no-extras --no-exact | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value %.0f ms %.4f seed1305 %.0f' % (d['value'], d['ms_per_step'], d['value_seed1305']), {k['stage']: round(k['avg_us'],1) for k in d['kernels']})"; }
for rep in 1 2 3; do
for v in lib_head lib_rotlanes; do
echo "== $v"; export ESAC_HIP_LIB=$GRAFT_REPO_ROOT/scratch/$v.so; run
done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/h_ab.txt
for v in lib_head; do
echo "== $v"; ESAC_HIP_LIB=$GRAFT_REPO_ROOT/scratch/$v.so python scripts/dev/host_turn.py 300 2>&1 | grep "launch\|per step"
done 2>&1 | tee gpurun_out/r05/h_turn.txt
ESAC_HIP_LIB=$GRAFT_REPO_ROOT/scratch/lib_rotlanes.so timeout 600 python scripts/dev/sweep.py 400 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r05/h_sweep.txt
