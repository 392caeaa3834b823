#!/bin/bash
# round 5, call B: the lane-dealt LM step -- GPU tests, a 600-frame sweep against the oracle, same-box A/B against round 4's library
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05/b_tests.txt
cat gpurun_out/r05/b_tests.txt
timeout 600 python scripts/dev/sweep.py 600 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/r05/b_sweep.txt
cat gpurun_out/r05/b_sweep.txt
bash scripts/dev/ab.sh scratch/lib_r04.so 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/b_ab.txt
cat gpurun_out/r05/b_ab.txt
