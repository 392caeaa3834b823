R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/${1:-r02e}
mkdir -p $O
bash scripts/dev/cyc.sh > $O/cyc.txt 2>&1; cat $O/cyc.txt
timeout 900 python bench.py --no-cpu-baseline --no-training --batch 0 > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -3 $O/bench_cfg2.err
python - $O/bench_cfg2.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.0f ms/step %.4f" % (d["value"], d["ms_per_step"]), d["phase_ms"])
for k in d["kernels"]: print(k["stage"], round(k["avg_us"],2), round(k["pct"],1))
print({k:v for k,v in d["roofline"].items() if k in ("achieved","frac","kernel_ms")}, d.get("with_h2d"))
PY
