#!/bin/bash
# round 4, call x: the team kernel after the register / instruction trims (single step site + LDS stash, accumulators
# reduced in place, DPP moves without a tied copy): parity tests, then same-box A/B against the committed build
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04x
timeout 900 python -m pytest tests/test_gpu_semantics.py tests/test_gpu_parity.py tests/test_gpu_backward.py tests/test_gpu_edge.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python scripts/dev/sweep.py 600 2>&1 | tail -3
bash scripts/dev/ab.sh scratch/lib_head.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04x/ab.txt
