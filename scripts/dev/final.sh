R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/final
timeout 600 python bench.py 2>gpurun_out/final/bench_cfg2.err | tail -1 > gpurun_out/final/bench_cfg2.json
timeout 300 python bench.py --experts 10 --hyps 1024 --steps 100 --warmup 10 --no-training --batch 0 2>/dev/null | tail -1 > gpurun_out/final/bench_cfg3.json
timeout 300 python bench.py --experts 50 --hyps 16384 --steps 20 --warmup 3 --no-cpu-baseline --no-training --batch 0 2>/dev/null | tail -1 > gpurun_out/final/bench_cfg5a.json
timeout 300 python bench.py --experts 50 --hyps 16384 --grid 480x640 --steps 5 --warmup 2 --no-cpu-baseline --no-training --batch 0 2>/dev/null | tail -1 > gpurun_out/final/bench_cfg5b.json
for b in 16 256; do timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-training --batch $b 2>/dev/null | tail -1 > gpurun_out/final/bench_batch$b.json; done
for f in cfg2 cfg3 cfg5a cfg5b batch16 batch256; do python - $f <<'PY'
import json,sys
d=json.load(open("gpurun_out/final/bench_%s.json"%sys.argv[1]))
print(sys.argv[1], "hyp/s %.0f ms/step %.4f" % (d["value"], d["ms_per_step"]), {k: round(v,4) for k,v in d["phase_ms"].items()}, "roofline GB/s", round(d["roofline"]["achieved"]), "kernel_ms", round(d["roofline"]["kernel_ms"],5), "cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("cores"), "batched", d.get("batched",{}).get("value"), "training", d.get("training",{}).get("ms_per_call"), d.get("training",{}).get("cpu_oracle_ms_per_call"))
PY
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/stats -o r01 -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-training --batch 0 > $R/gpurun_out/final/bench_under_rocprof.json 2> $R/gpurun_out/final/rocprof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/final/pmc_fetch -o r01 -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-training --batch 0 > /dev/null 2>> $R/gpurun_out/final/rocprof.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/final/pmc_write -o r01 -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-training --batch 0 > /dev/null 2>> $R/gpurun_out/final/rocprof.err
cd $R
python scripts/summarize_rocprof.py $(find gpurun_out/final/stats -name "*.db" | head -1) $(find gpurun_out/final/pmc_fetch -name "*.db" | head -1) $(find gpurun_out/final/pmc_write -name "*.db" | head -1) > gpurun_out/final/rocprof_summary.txt 2>&1
cat gpurun_out/final/rocprof_summary.txt
tail -1 gpurun_out/final/bench_under_rocprof.json | cut -c1-400
