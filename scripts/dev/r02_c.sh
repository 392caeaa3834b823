# VALU issue-rate micro-benchmark + SQ counters of the tiled score kernel on the 5b shape + default bench
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/${1:-r02c}
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/dev/valu_rate.hip -o /tmp/valu_rate 2>/dev/null && /tmp/valu_rate | tee $O/valu_rate.txt
cd /tmp && export TMPDIR=/tmp
B5="python $R/bench.py --experts 50 --hyps 16384 --grid 480x640 --steps 3 --warmup 1 --no-cpu-baseline --no-training --batch 0"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $R/$O/pmc5b_sq -o t -- $B5 > /dev/null 2> $R/$O/rocprof5b_sq.err
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $R/$O/pmc5b_tcc -o t -- $B5 > /dev/null 2> $R/$O/rocprof5b_tcc.err
cd $R
python - $O <<'PY'
import sqlite3, sys, glob
for sub in ("pmc5b_sq", "pmc5b_tcc"):
    for path in glob.glob("%s/%s/**/*.db" % (sys.argv[1], sub), recursive=True):
        d = sqlite3.connect(path)
        rows = d.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
        for k, c, n, v in rows:
            if "esac" in k:
                print("%-44s %-26s n=%d %.4g" % (k[:44], c, n, v))
PY
timeout 600 python bench.py --no-cpu-baseline --no-training --batch 0 | tail -1 | cut -c1-1500
