#!/bin/bash
# round 3, first GPU call: the whole GPU suite (incl. the new parity / robustness tests), then short bench lines
# usage (on the GPU box, from the repo root): bash scripts/dev/r03_first.sh
export TMPDIR=/tmp
O=gpurun_out/r03a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_cfg2_short.json 2> $O/bench_cfg2_short.err
timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench_cfg2.json 2> $O/bench_cfg2.err
for c in cfg3 cfg4 cfg5a cfg5b; do
  timeout 600 python bench.py --config $c --no-cpu-baseline --no-extras > $O/bench_$c.json 2> $O/bench_$c.err
done
python - <<'PY'
import json,glob
for p in sorted(glob.glob("gpurun_out/r03a/bench_*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1])
        print(p, "ms %.4f value %.0f" % (d["ms_per_step"], d["value"]), {k["stage"]: round(k["avg_us"],1) for k in d.get("kernels",[])})
    except Exception as e:
        print(p, "FAILED", e)
PY
