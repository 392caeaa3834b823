#!/bin/bash
# The soaks of a round on the final kernels (gpurun -- 'scripts/dev/soak_round.sh r06'): forward parity against the oracle over
# 3000 frames of four kinds + 1000 frames with several experts (the speculative route), the speculative route against the serial one,
# the training path over 300 calls -> gpurun_out/profiles_<round>/
RND=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
P=gpurun_out/profiles_$RND
mkdir -p $P
timeout 1500 python scripts/dev/sweep.py 3000 1000 77 2>&1 | grep -v amdgpu.ids | tail -5 > $P/${RND}_sweep_3000.txt
timeout 1500 python scripts/dev/sweep.py 1000 5000 1305 multi 2>&1 | grep -v amdgpu.ids | tail -6 > $P/${RND}_sweep_1000_several_experts.txt
timeout 900 python scripts/dev/spec_compare.py 40 2>&1 | grep -v amdgpu.ids | tail -8 > $P/${RND}_spec_vs_serial_240.txt
timeout 1500 python scripts/dev/bwd_sweep.py 300 2>&1 | grep -v amdgpu.ids | tail -8 > $P/${RND}_bwd_sweep_300.txt
cat $P/${RND}_sweep_3000.txt $P/${RND}_sweep_1000_several_experts.txt $P/${RND}_spec_vs_serial_240.txt $P/${RND}_bwd_sweep_300.txt
