#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_edge.py tests/test_gpu_parity.py tests/test_gpu_parity_large.py -m gpu -x -q 2>&1 | tail -4
bash scripts/dev/cyc.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/r05_refine_cycles_team8.txt
