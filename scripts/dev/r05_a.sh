#!/bin/bash
# round 5, call A: dependent-chain latency probe
set -x
mkdir -p gpurun_out/r05
cd /root/repo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value scripts/dev/lat_probe.hip -o /tmp/lat_probe && /tmp/lat_probe > gpurun_out/r05/lat_probe.txt 2>&1
cat gpurun_out/r05/lat_probe.txt
