#!/bin/bash
# round 5: rocprofv3 kernel trace of 60 esac_hip_backward calls on the final kernels (cfg2, alpha 100) -> profiles/r05_backward_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05bwd
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05bwd/stats -o r05 -- python $R/scripts/dev/bwd_loop.py 100 > $R/gpurun_out/r05bwd/run.txt 2> $R/gpurun_out/r05bwd/err.txt
cd $R
python scripts/summarize_rocprof.py $(find gpurun_out/r05bwd/stats -name "*.db" | head -1) > gpurun_out/r05bwd/r05_backward_cfg2_alpha100_rocprofv3_summary.txt 2>&1
cat gpurun_out/r05bwd/run.txt | tail -2
head -16 gpurun_out/r05bwd/r05_backward_cfg2_alpha100_rocprofv3_summary.txt
rm -rf gpurun_out/r05bwd/stats
