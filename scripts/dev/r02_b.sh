# round 2: GPU suite + profile of the 5b stress shape (kernel trace, FETCH_SIZE)
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/${1:-r02b}
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 > $O/pytest.log 2>&1
tail -60 $O/pytest.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
B5="python $R/bench.py --experts 50 --hyps 16384 --grid 480x640 --steps 3 --warmup 1 --no-cpu-baseline --no-training --batch 0"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/stats5b -o t -- $B5 > $R/$O/bench5b_under_rocprof.json 2> $R/$O/rocprof5b.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/pmc5b_fetch -o t -- $B5 > /dev/null 2>> $R/$O/rocprof5b.err
cd $R
python scripts/summarize_rocprof.py $(find $O/stats5b -name "*.db" | head -1) $(find $O/pmc5b_fetch -name "*.db" | head -1) > $O/cfg5b_summary.txt 2>&1
cat $O/cfg5b_summary.txt
tail -1 $O/bench5b_under_rocprof.json | cut -c1-600
