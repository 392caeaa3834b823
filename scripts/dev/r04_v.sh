#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04v
ESAC_SLOT_TEAMS=0 timeout 1500 python scripts/dev/bwd_sweep.py 300 > gpurun_out/r04v/r04_bwd_sweep_300_one_workgroup.txt 2>&1
grep -c MISMATCH gpurun_out/r04v/r04_bwd_sweep_300_one_workgroup.txt
grep "MISMATCH frame 181\|MISMATCH frame 196\|MISMATCH frame 232\|backward calls" gpurun_out/r04v/r04_bwd_sweep_300_one_workgroup.txt
ESAC_HIP_LIB=$PWD/scratch/lib_old.so timeout 1500 python scripts/dev/bwd_sweep.py 300 2>&1 | grep "MISMATCH frame 181\|MISMATCH frame 196\|MISMATCH frame 232\|backward calls"
