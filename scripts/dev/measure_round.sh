#!/bin/bash
# The measurement set of a round (on the GPU box: gpurun -- 'scripts/dev/measure_round.sh r06'): bench lines of every BASELINE
# workload (with CPU baseline and accuracy block), the driver-style 20-step line, batch sizes, then per workload the rocprofv3
# kernel trace + PMC passes (profile_cfg.sh) -> gpurun_out/profiles_<round>/ (copy what is to be judged into profiles/).
# BENCH_ONLY=1 skips the rocprofv3 passes; CFGS="cfg2 cfg3" restricts the workloads.
RND=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
P=gpurun_out/profiles_$RND
mkdir -p $P
CFGS=${CFGS:-"cfg2 cfg3 cfg4 cfg5a cfg5b"}
last() { grep '^{' | tail -1; }
declare -A ARGS=( [cfg2]="" [cfg3]="--steps 200 --warmup 20 --no-extras" [cfg4]="--steps 100 --warmup 10 --no-extras --no-exact"
                  [cfg5a]="--steps 30 --warmup 3 --no-extras" [cfg5b]="--steps 8 --warmup 2 --no-extras" )
declare -A PARGS=( [cfg2]="" [cfg3]="--steps 100 --warmup 10" [cfg4]="--steps 60 --warmup 6" [cfg5a]="--steps 12 --warmup 2" [cfg5b]="--steps 4 --warmup 1" )
# the rocprofv3 passes FIRST: the bench lines then embed the profiles of the tree they run on (profile_stale: false) -- bench.py reads
# the newest profiles/r*_<cfg>_kernels.json, so the summaries are copied there on the box before the lines are taken
if [ -z "$BENCH_ONLY" ]; then
  for c in $CFGS; do bash scripts/dev/profile_cfg.sh $c $RND ${PARGS[$c]} > $P/log_$c.txt 2>&1; done
  cp $P/${RND}_cfg*_kernels.json $P/${RND}_cfg*_kernels.txt profiles/ 2>/dev/null
  rm -rf gpurun_out/prof_${RND}_*
fi
for c in $CFGS; do
  timeout 1200 python bench.py --config $c ${ARGS[$c]} 2> $P/err_$c.txt | last > $P/${RND}_bench_$c.json
done
if [[ " $CFGS " == *" cfg2 "* ]]; then
  timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | last > $P/${RND}_bench_cfg2_driver_style.json
  for b in 16 32 256; do timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-training --no-exact --batch $b 2>/dev/null | last > $P/${RND}_bench_batch$b.json; done
fi
for f in $P/${RND}_bench_*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("bench_")[-1][:-5], "hyp/s %.0f ms/step %.4f" % (d["value"], d["ms_per_step"]), [(k["stage"], round(k["avg_us"], 1)) for k in d.get("kernels", [])],
          "roofline", d["roofline"]["bound"][:12], round(d["roofline"]["frac"], 3), "cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"),
          "acc", (d.get("accuracy") or {}).get("median_rot_err_rad"), (d.get("accuracy") or {}).get("winner_match"), "batched", d.get("batched", {}).get("value"),
          "training", d.get("training", {}).get("ms_per_call"), "h2d", d.get("with_h2d", {}).get("value"), "seed1305", d.get("value_seed1305"),
          "exact", (d.get("value_exact") or {}).get("value"), "fast", (d.get("value_fast") or {}).get("value"), "sharded1", (d.get("sharded_world1") or {}).get("overhead_us"),
          "spec", d.get("speculation"), "host", (d.get("host_turn_us") or {}).get("split_us"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
if [ -z "$BENCH_ONLY" ]; then cat $P/${RND}_cfg*_kernels.txt; fi
