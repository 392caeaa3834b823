#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
run() { timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-extras --no-exact | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value %.0f ms %.4f seed1305 %.0f' % (d['value'], d['ms_per_step'], d['value_seed1305']), {k['stage']: round(k['avg_us'],1) for k in d['kernels']})"; }
for rep in 1 2; do
for v in lib_head lib_tree lib_dummy lib_teamfirst; do
echo "== $v"; export ESAC_HIP_LIB=$GRAFT_REPO_ROOT/scratch/$v.so; run
done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/j_layout.txt
