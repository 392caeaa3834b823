# round 2, first GPU session: the whole GPU suite on the new ABI + baseline profile of the 5b stress shape
R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/r02a
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --maxfail=40 -x --deselect tests/test_gpu_semantics.py::test_rank_deficient_refit_follows_the_svd_route > $O/pytest.log 2>&1
tail -40 $O/pytest.log
timeout 300 python -m pytest tests/test_gpu_semantics.py -m gpu -q -k rank_deficient > $O/pytest_rank.log 2>&1
tail -30 $O/pytest_rank.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/$O/counters.txt 2>&1
B5="python $R/bench.py --experts 50 --hyps 16384 --grid 480x640 --steps 3 --warmup 1 --no-cpu-baseline --no-training --batch 0"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/stats5b -o base -- $B5 > $R/$O/bench5b_under_rocprof.json 2> $R/$O/rocprof5b.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/pmc5b_fetch -o base -- $B5 > /dev/null 2>> $R/$O/rocprof5b.err
cd $R
python scripts/summarize_rocprof.py $(find $O/stats5b -name "*.db" | head -1) $(find $O/pmc5b_fetch -name "*.db" | head -1) > $O/cfg5b_baseline_summary.txt 2>&1
cat $O/cfg5b_baseline_summary.txt
