#!/bin/bash
# round 4: first run of the team refinement -- parity subset, then the headline bench for team sizes 0 / 2 / 4 / 8
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r04b
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py -m gpu -x -q > $O/pytest.log 2>&1
tail -15 $O/pytest.log
for g in 0 2 4 8 0 8; do
  ESAC_REFINE_TEAM=$g timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench_team$g.json 2> $O/bench_team$g.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_team$g.json").read().strip().splitlines()[-1])
    print("team $g: ms %.4f value %.0f seed1305 %s" % (d["ms_per_step"], d["value"], d.get("value_seed1305")), {k["stage"]: round(k["avg_us"],1) for k in d.get("kernels",[])})
except Exception as e:
    print("team $g FAILED", e); print(open("$O/bench_team$g.err").read()[-2000:])
PY
done
ESAC_REFINE_TEAM=8 bash scripts/dev/cyc.sh > $O/cyc_team8.txt 2>&1
cat $O/cyc_team8.txt
