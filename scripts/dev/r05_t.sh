#!/bin/bash
# round 5, call T: what the driver runs at round end, on the final tree: smoke(), the default bench line (JSON = last stdout line, wall time)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
t0=$(date +%s.%N)
timeout 900 python bench.py > gpurun_out/r05/t_bench_stdout.txt 2> gpurun_out/r05/t_bench_stderr.txt
echo "bench.py default: $(echo "$(date +%s.%N) - $t0" | bc) s wall, rc $?"
python - <<'PY'
import json
lines = open("gpurun_out/r05/t_bench_stdout.txt").read().splitlines()
d = json.loads(lines[-1])
print("stdout lines", len(lines), "| last line is the JSON: value %.0f %s, ms_per_step %.4f, steps %d, roofline frac %.3f, cpu_baseline %.0f (%s, %d cores), profile_stale %s" % (
    d["value"], d["unit"], d["ms_per_step"], d["steps"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], d["cpu_baseline"]["cores"], d.get("profile_stale")))
PY
