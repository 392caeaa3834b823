#!/bin/bash
# round 5, soak of the final kernels: 3000-frame forward sweep and 300 backward calls against the oracle
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05soak
timeout 900 python scripts/dev/sweep.py 3000 2>&1 | grep -v amdgpu.ids | tail -5 > gpurun_out/r05soak/r05_sweep_3000.txt
cat gpurun_out/r05soak/r05_sweep_3000.txt
timeout 600 python scripts/dev/bwd_sweep.py 300 2>&1 | grep -v amdgpu.ids > gpurun_out/r05soak/r05_bwd_sweep_300.txt
tail -3 gpurun_out/r05soak/r05_bwd_sweep_300.txt
bash scripts/dev/cyc.sh 2>&1 | grep -v amdgpu.ids > gpurun_out/r05soak/r05_refine_cycles_team8.txt
tail -8 gpurun_out/r05soak/r05_refine_cycles_team8.txt
