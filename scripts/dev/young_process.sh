#!/bin/bash
# per-call kernel durations of the first 80 identical calls of a fresh process (rocprofv3 kernel trace) -> gpurun_out/<name>.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/yp
mkdir -p $O
cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $O/tr -o t -- python $R/scripts/dev/young_process.py > $O/run.txt 2> $O/err.txt
cd $R
python - $(find $O/tr -name "*.db" | head -1) <<'PY' | tee gpurun_out/${1:-young_process}.txt
import sqlite3, sys, collections
import numpy as np
db = sqlite3.connect(sys.argv[1])
rows = [r for r in db.execute("select name, start, end from kernels order by start") if "esac::" in r[0]]
per = collections.OrderedDict()
for n, s, e in rows:
    per.setdefault(n.replace("void esac::", "").split("(")[0], []).append((e - s) / 1e3)
for k, v in per.items():
    v = np.array(v)
    print("%-28s n %3d | calls 0-4 %6.2f | 5-24 %6.2f | 25-44 %6.2f | 45-79 %6.2f us (same work every call)" % (k, len(v), v[:5].mean(), v[5:25].mean(), v[25:45].mean(), v[45:].mean()))
PY
cat $O/run.txt | tail -9
rm -rf $O/tr
