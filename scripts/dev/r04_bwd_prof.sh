#!/bin/bash
# round 4: rocprofv3 kernel trace of 60 esac_hip_backward calls on the final kernels (cfg2, alpha 100) -> profiles/r04_backward_*
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04bwd
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04bwd/stats -o r04 -- python $R/scripts/dev/bwd_loop.py 100 > $R/gpurun_out/r04bwd/run.txt 2> $R/gpurun_out/r04bwd/err.txt
cd $R
python scripts/summarize_rocprof.py $(find gpurun_out/r04bwd/stats -name "*.db" | head -1) > gpurun_out/r04bwd/r04_backward_cfg2_alpha100_rocprofv3_summary.txt 2>&1
cat gpurun_out/r04bwd/run.txt | tail -2
head -16 gpurun_out/r04bwd/r04_backward_cfg2_alpha100_rocprofv3_summary.txt
rm -rf gpurun_out/r04bwd/stats
