# round 5 measurement set -> gpurun_out/profiles_r05 (copied to profiles/ afterwards); BENCH_ONLY=1 skips the rocprofv3 passes
R=$GRAFT_REPO_ROOT
cd $R
export TMPDIR=/tmp
P=gpurun_out/profiles_r05
mkdir -p $P
last() { grep '^{' | tail -1; }
timeout 900 python bench.py 2> $P/err_cfg2.txt | last > $P/r05_bench_cfg2.json
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | last > $P/r05_bench_cfg2_driver_style.json
timeout 600 python bench.py --config cfg3 --steps 200 --warmup 20 --no-extras 2>/dev/null | last > $P/r05_bench_cfg3.json
timeout 600 python bench.py --config cfg4 --steps 100 --warmup 10 --no-extras --no-exact 2>/dev/null | last > $P/r05_bench_cfg4.json
timeout 900 python bench.py --config cfg5a --steps 30 --warmup 3 --no-extras 2>/dev/null | last > $P/r05_bench_cfg5a.json
timeout 1200 python bench.py --config cfg5b --steps 8 --warmup 2 --no-extras 2>/dev/null | last > $P/r05_bench_cfg5b.json
for b in 16 32 256; do timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-training --no-exact --batch $b 2>/dev/null | last > $P/r05_bench_batch$b.json; done
python scripts/dev/host_turn.py 400 2>&1 | grep -v amdgpu.ids > $P/r05_host_turn_python.txt
tests/native/abi_check esac_amd/libesac_hip.so gpu time 2>&1 | grep "host turn" > $P/r05_host_turn_c.txt
for f in cfg2 cfg2_driver_style cfg3 cfg4 cfg5a cfg5b batch16 batch32 batch256; do python - $P/r05_bench_$f.json $f <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[2], "hyp/s %.0f ms/step %.4f" % (d["value"], d["ms_per_step"]), [(k["stage"], round(k["avg_us"],1)) for k in d.get("kernels",[])], "roofline", d["roofline"]["bound"][:12], round(d["roofline"]["frac"],3), "cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"), "acc", (d.get("accuracy") or {}).get("median_rot_err_rad"), (d.get("accuracy") or {}).get("winner_match"), "batched", d.get("batched",{}).get("value"), "training", d.get("training",{}).get("ms_per_call"), "h2d", d.get("with_h2d",{}).get("value"), "seed1305", d.get("value_seed1305"), "exact", (d.get("value_exact") or {}).get("value"), "fast", (d.get("value_fast") or {}).get("value"), "sharded1", (d.get("sharded_world1") or {}).get("overhead_us"), "refine", (d.get("refine") or {}).get("mode"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
if [ -z "$BENCH_ONLY" ]; then
bash scripts/dev/profile_cfg.sh cfg2 r05 > $P/log_cfg2.txt 2>&1
bash scripts/dev/profile_cfg.sh cfg3 r05 --steps 100 --warmup 10 > $P/log_cfg3.txt 2>&1
bash scripts/dev/profile_cfg.sh cfg4 r05 --steps 60 --warmup 6 > $P/log_cfg4.txt 2>&1
bash scripts/dev/profile_cfg.sh cfg5a r05 --steps 12 --warmup 2 > $P/log_cfg5a.txt 2>&1
bash scripts/dev/profile_cfg.sh cfg5b r05 --steps 4 --warmup 1 > $P/log_cfg5b.txt 2>&1
cat $P/r05_cfg*_kernels.txt
rm -rf gpurun_out/prof_r05_*
fi
