#!/bin/bash
# round 5, call N: the first 80 calls of a fresh process, round 4's tree against this one, call by call (same frames, same keys)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
echo "== r04"; GRAFT_REPO_ROOT=$PWD/scratch/r04tree python scratch/first_calls_r04.py 2>&1 | grep -v amdgpu.ids | cut -c1-95
echo "== r05 (fast route: flags 0)"; ESAC_FC_FAST=1 python scripts/dev/first_calls.py 2>&1 | grep -v amdgpu.ids | cut -c1-95
echo "== r05 (auto exact)"; python scripts/dev/first_calls.py 2>&1 | grep -v amdgpu.ids | cut -c1-95
done
