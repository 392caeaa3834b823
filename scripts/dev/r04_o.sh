#!/bin/bash
# team sizes beyond 8: the GPU tests of the team, then the headline bench for 8 / 10 / 12 / 16 / 19 members, twice
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04o
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_semantics.py tests/test_gpu_parity.py tests/test_gpu_edge.py -m gpu -q -x > $O/pytest.log 2>&1
tail -15 $O/pytest.log
for rep in 1 2; do
for g in 8 10 12 16 19 24; do
  ESAC_REFINE_TEAM=$g timeout 300 python bench.py --no-cpu-baseline --no-extras --no-exact > $O/bench_team$g.json 2> $O/bench_team$g.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_team$g.json").read().strip().splitlines()[-1])
    print("team $g: ms %.4f value %.0f seed1305 %s" % (d["ms_per_step"], d["value"], d.get("value_seed1305")), {k["stage"]: round(k["avg_us"],1) for k in d.get("kernels",[])}, d.get("refine",{}).get("workgroups"))
except Exception as e:
    print("team $g FAILED", e); print(open("$O/bench_team$g.err").read()[-2000:])
PY
done
done
