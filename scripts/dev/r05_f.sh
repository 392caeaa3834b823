#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value scripts/dev/launch_cost.hip -o /tmp/launch_cost && /tmp/launch_cost 2>&1 | tee gpurun_out/launch_cost.txt
echo "== default stream"; python scripts/dev/host_turn.py 300 2>&1 | grep -v amdgpu.ids | tee gpurun_out/host_turn_default.txt
echo "== side stream"; python scripts/dev/side_stream_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/host_turn_side.txt
