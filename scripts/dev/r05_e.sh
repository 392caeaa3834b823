#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python scripts/dev/stall_probe.py 2>&1 | grep -v amdgpu.ids
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_stall -o stall -- python $GRAFT_REPO_ROOT/scripts/dev/stall_probe.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
for fn in glob.glob('/tmp/prof_stall/**/*kernel_trace.csv', recursive=True):
    rows = list(csv.DictReader(open(fn)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    t0 = int(rows[0]['Start_Timestamp'])
    for r in rows[-40:]:
        print('%10.1f us  dur %9.1f us  %s  grid %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:60], r.get('Grid_Size_X', r.get('Grid_Size'))))
PY
