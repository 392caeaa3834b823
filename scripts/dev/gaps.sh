#!/bin/bash
# where the headline call's time goes between its kernels: rocprofv3 kernel trace of the default bench, start / end stamps
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/gaps
mkdir -p $O
cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $O/tr -o t -- python $R/bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-extras --no-exact > $O/bench.json 2> $O/err.txt
python - $(find $O/tr -name "*.db" | head -1) <<'PY'
import sqlite3, sys, numpy as np
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else None
cols = [r[1] for r in db.execute("pragma table_info(%s)" % view)]
print(cols)
rows = list(db.execute("select name, start, end from %s order by start" % view))
rows = [r for r in rows if "esac::" in r[0]]
names = [r[0].replace("void esac::","").split("(")[0][:30] for r in rows]
st = np.array([r[1] for r in rows], float); en = np.array([r[2] for r in rows], float)
# steady state: last 60 %
k0 = int(len(rows) * 0.4)
import collections
gap = collections.defaultdict(list); dur = collections.defaultdict(list)
for i in range(k0, len(rows) - 1):
    dur[names[i]].append((en[i] - st[i]) / 1e3)
    gap[names[i] + " -> " + names[i + 1]].append((st[i + 1] - en[i]) / 1e3)
for k, v in dur.items(): print("kernel %-32s n %4d  mean %7.2f us  median %7.2f" % (k, len(v), np.mean(v), np.median(v)))
for k, v in gap.items(): print("gap    %-64s n %4d  mean %7.2f us  median %7.2f" % (k, len(v), np.mean(v), np.median(v)))
PY
tail -1 $O/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('under rocprof: ms_per_step %.4f' % d['ms_per_step'])"
rm -rf $O/tr
