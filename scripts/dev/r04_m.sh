#!/bin/bash
# the device-side campaign of the sampling screen: 4e10 tries per map, both families (~1e12 tries)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m
mkdir -p $O
timeout 1500 python scripts/dev/screen_campaign_device.py 4e10 both > $O/r04_screen_campaign_device.txt 2>&1
tail -3 $O/r04_screen_campaign_device.txt
