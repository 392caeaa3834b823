#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r05/d_tests.txt
cat gpurun_out/r05/d_tests.txt
