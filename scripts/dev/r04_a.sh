#!/bin/bash
# round 4, first GPU call: exchange micro-benchmark (placement + tagged-granule latency), refinement cycle breakdown,
# baseline bench of the round-3 kernels on this box
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r04a
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/dev/xcd_exchange.hip -o /tmp/xcd 2>/dev/null
timeout 120 /tmp/xcd > $O/xcd_exchange.txt 2>&1
cat $O/xcd_exchange.txt
bash scripts/dev/cyc.sh > $O/cyc.txt 2>&1
cat $O/cyc.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_cfg2_short.json 2> $O/bench_cfg2_short.err
timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python - <<'PY'
import json,glob
for p in sorted(glob.glob("gpurun_out/r04a/bench_*.json")):
    try:
        d=json.loads(open(p).read().strip().splitlines()[-1])
        print(p, "ms %.4f value %.0f seed1305 %s" % (d["ms_per_step"], d["value"], d.get("value_seed1305")), {k["stage"]: round(k["avg_us"],1) for k in d.get("kernels",[])})
    except Exception as e:
        print(p, "FAILED", e)
PY
