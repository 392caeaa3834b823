#!/bin/bash
# per-kernel table of the batched extra of bench.py (frames per launch = $1): kernel instances ranked by their longest
# calls, so the batch launches stand out from the single-frame ones
R=$GRAFT_REPO_ROOT
B=${1:-256}
O=$R/gpurun_out/kstats_batch$B
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats -o t -- python $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-training --batch $B > $O/bench.json 2> $O/err.txt
cd $R
python - $(find $O/stats -name "*.db" | head -1) <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
t = [x for x in tabs if "kernel_dispatch" in x][0]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % t)]
print(t, cols[:20])
PY
