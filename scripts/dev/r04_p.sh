#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04p
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r04p/pytest.log 2>&1
tail -6 gpurun_out/r04p/pytest.log
bash scripts/dev/r04_final.sh > gpurun_out/r04p/final.log 2>&1
grep -v "^$" gpurun_out/r04p/final.log | head -60
