#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -12 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_short.json 2> $O/bench_short.err
timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python - <<'PY'
import json
for n in ("bench_short", "bench_cfg2"):
    try:
        d = json.loads(open("gpurun_out/r04k/%s.json" % n).read().strip().splitlines()[-1])
        print(n, "ms %.4f value %.0f seed1305 %s exact %s" % (d["ms_per_step"], d["value"], d.get("value_seed1305"), d.get("value_exact", {}).get("value")),
              {k["stage"]: round(k["avg_us"], 1) for k in d.get("kernels", [])}, d.get("refine"))
    except Exception as e:
        print(n, "FAILED", e, open("gpurun_out/r04k/%s.err" % n).read()[-1500:])
PY
