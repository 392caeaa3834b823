# the round's measurement set: bench lines + rocprofv3 / PMC profiles for configs 2, 3, 5a, 5b (-> gpurun_out/profiles_r02, copied to profiles/)
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/profiles_r02
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/dev/valu_rate.hip -o /tmp/valu_rate 2>/dev/null && /tmp/valu_rate > gpurun_out/profiles_r02/r02_valu_rate.txt
bash scripts/dev/profile_cfg.sh cfg2 r02 > gpurun_out/profiles_r02/log_cfg2.txt 2>&1
bash scripts/dev/profile_cfg.sh cfg3 r02 --steps 100 --warmup 10 > gpurun_out/profiles_r02/log_cfg3.txt 2>&1
bash scripts/dev/profile_cfg.sh cfg5a r02 --steps 12 --warmup 2 > gpurun_out/profiles_r02/log_cfg5a.txt 2>&1
bash scripts/dev/profile_cfg.sh cfg5b r02 --steps 4 --warmup 1 > gpurun_out/profiles_r02/log_cfg5b.txt 2>&1
cat gpurun_out/profiles_r02/r02_cfg*_kernels.txt
