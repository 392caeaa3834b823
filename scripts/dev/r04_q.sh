#!/bin/bash
# same-box A/B of two builds of the library: scratch/lib_old.so (the committed tree) against the working tree
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04q
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_semantics.py tests/test_gpu_edge.py -m gpu -q -x > $O/pytest.log 2>&1
tail -4 $O/pytest.log
for rep in 1 2 3; do
for v in old new; do
  if [ $v = old ]; then export ESAC_HIP_LIB=$PWD/scratch/lib_old.so; else unset ESAC_HIP_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline --no-extras --no-exact > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
    print("$v: ms %.4f value %.0f seed1305 %s" % (d["ms_per_step"], d["value"], d.get("value_seed1305")), {k["stage"]: round(k["avg_us"],1) for k in d.get("kernels",[])})
except Exception as e:
    print("$v FAILED", e); print(open("$O/bench_$v.err").read()[-2000:])
PY
done
done
unset ESAC_HIP_LIB
ESAC_REFINE_TEAM=8 bash scripts/dev/cyc.sh 2>&1 | grep -v "^-\|ep_\|amdgpu.ids"
