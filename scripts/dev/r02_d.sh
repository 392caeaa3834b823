R=$GRAFT_REPO_ROOT
cd $R
O=gpurun_out/${1:-r02d}
mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q --maxfail=20 > $O/pytest.log 2>&1
tail -25 $O/pytest.log | cut -c1-250
timeout 900 python bench.py > $O/bench_cfg2.json 2> $O/bench_cfg2.err; tail -3 $O/bench_cfg2.err; cut -c1-3000 $O/bench_cfg2.json
bash scripts/dev/profile_cfg.sh cfg2 r02 > $O/prof_cfg2.log 2>&1; tail -12 $O/prof_cfg2.log
bash scripts/dev/profile_cfg.sh cfg5b r02 --steps 3 --warmup 1 > $O/prof_cfg5b.log 2>&1; tail -14 $O/prof_cfg5b.log
