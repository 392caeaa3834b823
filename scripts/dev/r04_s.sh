#!/bin/bash
# one-workgroup refinement with the team kernel's serial-section trims: whole GPU suite, then old / new library on one box
# (headline in one-workgroup mode, batches of 16 / 64 frames, backward)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04s
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log
for rep in 1 2; do
for v in old new; do
  if [ $v = old ]; then export ESAC_HIP_LIB=$PWD/scratch/lib_old.so; else unset ESAC_HIP_LIB; fi
  ESAC_REFINE_TEAM=0 timeout 300 python bench.py --no-cpu-baseline --no-exact --batch 16 > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
    print("$v (one workgroup): ms %.4f value %.0f" % (d["ms_per_step"], d["value"]), {k["stage"]: round(k["avg_us"],1) for k in d.get("kernels",[])}, "batch16 %.2f M" % (d["batched"]["value"]/1e6), "backward %.4f ms" % d["training"]["ms_per_call"])
except Exception as e:
    print("$v FAILED", e); print(open("$O/bench_$v.err").read()[-2000:])
PY
done
done
