# the round's last GPU call in ONE go: the whole GPU suite, then the profiles, copied into profiles/ on the box so that the bench
# lines that follow embed the profiles of the very kernels they run (profile_stale false)
R=$GRAFT_REPO_ROOT
cd $R
export TMPDIR=/tmp
P=gpurun_out/profiles_r04
mkdir -p $P
timeout 1800 python -m pytest tests -m gpu -q > $P/pytest_gpu.log 2>&1
tail -3 $P/pytest_gpu.log
bash scripts/dev/profile_cfg.sh cfg2 r04 > $P/log_cfg2.txt 2>&1
bash scripts/dev/profile_cfg.sh cfg3 r04 --steps 100 --warmup 10 > $P/log_cfg3.txt 2>&1
bash scripts/dev/profile_cfg.sh cfg4 r04 --steps 60 --warmup 6 > $P/log_cfg4.txt 2>&1
bash scripts/dev/profile_cfg.sh cfg5a r04 --steps 12 --warmup 2 > $P/log_cfg5a.txt 2>&1
bash scripts/dev/profile_cfg.sh cfg5b r04 --steps 4 --warmup 1 > $P/log_cfg5b.txt 2>&1
rm -rf gpurun_out/prof_r04_*
cp $P/r04_cfg*_kernels.json profiles/
BENCH_ONLY=1 bash scripts/dev/r04_final.sh
bash scripts/dev/gaps.sh > $P/r04_cfg2_gaps.txt 2>&1
ESAC_REFINE_TEAM=8 bash scripts/dev/cyc.sh > $P/r04_refine_cycles_team8.txt 2>&1
