#!/bin/bash
# quick per-kernel table of one bench.py workload (rocprofv3 --kernel-trace --stats only, no counters)
# usage (GPU box): bash scripts/dev/kstats.sh <cfg> [extra bench args]
R=$GRAFT_REPO_ROOT
CFG=$1; shift
O=$R/gpurun_out/kstats_$CFG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o t -- python $R/bench.py --config $CFG --no-cpu-baseline --no-extras $* > $O/bench.json 2> $O/err.txt
cd $R
python - $(find $O/stats -name "*.db" | head -1) $CFG <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = [r for r in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels") if "esac::" in r[0]]
steps = max([r[1] for r in rows if "k_refine" in r[0]] + [1])
print("== %s: %d calls traced" % (sys.argv[2], steps))
for name, calls, total, avg, pct in sorted(rows, key=lambda r: -r[2]):
    print("%-52s calls %5d avg %9.2f us  per call %9.2f us  %5.1f%%" % (name.replace("void ", "").replace("(esac::KArgs)", "")[:52], calls, avg / 1e3 if avg > 1e5 else avg, total / steps / (1e3 if avg > 1e5 else 1), pct))
PY
rm -rf $O/stats
