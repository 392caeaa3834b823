#!/bin/bash
# round 5, call S: granules of a one-XCD team stay in L2 (plain stores after the census) -- placement probe again, same-box A/B
# against the previous commit (several processes each: the old state is a per-process lottery), full GPU suite, 600-frame sweep
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
echo "== placement probe, new kernels"
ESAC_HIP_LIB=$GRAFT_REPO_ROOT/scratch/lib_granshift2.so timeout 300 python scripts/dev/gran_shift_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/s_gran_shift_local.txt | sed -n '1,8p;19,30p'
run() { timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-extras --no-exact | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('value %.0f ms %.4f seed1305 %.0f' % (d['value'], d['ms_per_step'], d['value_seed1305']), {k['stage']: round(k['avg_us'],1) for k in d['kernels']}, d['refine']['xcd_census'])"; }
for rep in 1 2 3 4; do
for v in head tree; do
if [ $v = head ]; then export ESAC_HIP_LIB=$GRAFT_REPO_ROOT/scratch/lib_head.so; else unset ESAC_HIP_LIB; fi
echo "== $v: $(run 2>&1 | grep -v amdgpu.ids | tail -1)"
done
done | tee gpurun_out/r05/s_ab.txt
unset ESAC_HIP_LIB
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05/s_tests.txt 2>&1
grep -a "passed\|failed\|error" gpurun_out/r05/s_tests.txt | tail -3
timeout 600 python scripts/dev/sweep.py 600 2>&1 | grep -v amdgpu.ids | tail -2 | tee gpurun_out/r05/s_sweep.txt
