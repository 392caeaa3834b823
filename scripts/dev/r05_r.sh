#!/bin/bash
# round 5, call R: the full GPU suite on the final tree (result to a file: RCCL's banner lands after pytest's summary on stdout),
# then the granule-placement probe in three processes
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r05/r_tests.txt 2>&1
grep -a "passed\|failed\|error" gpurun_out/r05/r_tests.txt | tail -3
for rep in 1 2 3; do
echo "== process $rep"
ESAC_HIP_LIB=$GRAFT_REPO_ROOT/scratch/lib_granshift.so timeout 300 python scripts/dev/gran_shift_probe.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r05/r_gran_shift.txt
