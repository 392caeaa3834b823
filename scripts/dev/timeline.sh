#!/bin/bash
# usage (on the GPU box): scripts/dev/timeline.sh <out-name> <python script + args ...>
# rocprofv3 kernel trace of the command; then per call (a call starts at a sampling kernel) the mean start offset, duration and
# queue of every kernel, the gaps between consecutive kernels on the critical path, and the mean call period -- steady state
# (last 60 % of the calls).  Output: gpurun_out/<out-name>.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; shift
O=$R/gpurun_out/tl_$NAME
mkdir -p $O
cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $O/tr -o t -- python $R/"$@" > $O/run.txt 2> $O/err.txt
cd $R
python - $(find $O/tr -name "*.db" | head -1) > gpurun_out/$NAME.txt <<'PY'
import sqlite3, sys, collections
import numpy as np
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(db.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")))
def short(n):
    return n.replace("void esac::", "").replace("esac::", "").split("(")[0][:44]
calls, cur, prev = [], None, ""
for r in rows:
    n = short(r[0])
    if "esac" not in r[0] and cur is None:
        continue
    start = n.startswith(("k_pack_cells", "k_sample_first", "k_sample<")) or (n.startswith("k_pending_list") and not prev.startswith("k_sample_first"))
    if start and prev.startswith("k_pack_cells"):
        start = False
    if start:
        cur = []
        calls.append(cur)
    if cur is not None:
        cur.append((n, r[1], r[2], r[3] if qcol else 0))
    prev = n
calls = calls[int(len(calls) * 0.4):-1]
print("calls analysed: %d (columns: %s)" % (len(calls), qcol))
per = collections.OrderedDict()
for c in calls:
    t0 = c[0][1]
    seen = collections.Counter()
    for n, s, e, q in c:
        seen[n] += 1
        key = "%s#%d" % (n, seen[n]) if seen[n] > 1 else n
        per.setdefault(key, []).append(((s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, q))
print("%-48s %5s %9s %9s %9s %8s %8s %8s  %s" % ("kernel", "n", "start_us", "dur_us", "end_us", "dur_min", "dur_med", "dur_max", "queue(s)"))
for k, v in per.items():
    a = np.array([x[:3] for x in v])
    print("%-48s %5d %9.2f %9.2f %9.2f %8.2f %8.2f %8.2f  %s" % (k, len(v), a[:, 0].mean(), a[:, 1].mean(), a[:, 2].mean(), a[:, 1].min(), np.median(a[:, 1]), a[:, 1].max(),
                                                            sorted(set(x[3] for x in v))))
ends = [max(e for _, _, e, _ in c) - c[0][1] for c in calls]
period = [b[0][1] - a[0][1] for a, b in zip(calls[:-1], calls[1:])]
busy = [sum(e - s for _, s, e, _ in c) for c in calls]
print("first kernel start -> last kernel end: mean %.2f us; call period: mean %.2f us, median %.2f us; sum of kernel durations: %.2f us"
      % (np.mean(ends) / 1e3, np.mean(period) / 1e3, np.median(period) / 1e3, np.mean(busy) / 1e3))
PY
tail -3 $O/run.txt
cat gpurun_out/$NAME.txt
rm -rf $O/tr
