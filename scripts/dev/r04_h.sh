#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1
tail -25 $O/pytest.log
for g in 8 0 8; do
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
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_short.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_short.json').read().strip().splitlines()[-1]); print('driver-style: ms %.4f value %.0f seed1305 %s' % (d['ms_per_step'], d['value'], d.get('value_seed1305')))"
