# the round's last GPU call in ONE go: profiles first, copied into profiles/ on the box so that the bench lines that follow embed
# the profiles of the very kernels they run (profile_stale false), then a test subset around the sampling screen
R=$GRAFT_REPO_ROOT
cd $R
P=gpurun_out/profiles_r03
mkdir -p $P
bash scripts/dev/profile_cfg.sh cfg2 r03 > $P/log_cfg2.txt 2>&1
bash scripts/dev/profile_cfg.sh cfg3 r03 --steps 100 --warmup 10 > $P/log_cfg3.txt 2>&1
bash scripts/dev/profile_cfg.sh cfg4 r03 --steps 60 --warmup 6 > $P/log_cfg4.txt 2>&1
bash scripts/dev/profile_cfg.sh cfg5a r03 --steps 12 --warmup 2 > $P/log_cfg5a.txt 2>&1
bash scripts/dev/profile_cfg.sh cfg5b r03 --steps 4 --warmup 1 > $P/log_cfg5b.txt 2>&1
rm -rf gpurun_out/prof_r03_*
cp $P/r03_cfg*_kernels.json profiles/
BENCH_ONLY=1 bash scripts/dev/r03_final.sh
timeout 240 python -m pytest tests -m gpu -x -q -k "adversarial or config3 or config4 or config5a or exact_sampling or two_phase" 2>&1 | grep -E "passed|failed|error" | tail -3
