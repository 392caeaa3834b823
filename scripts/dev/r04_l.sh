#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_semantics.py -m gpu -q -k "one_sided or adversarial" > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 600 python scripts/dev/screen_campaign_device.py 4e8 both > $O/campaign_rate.txt 2>&1
tail -4 $O/campaign_rate.txt
