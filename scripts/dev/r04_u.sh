#!/bin/bash
# soak: forward parity of the team route against the oracle over 3000 frames of four kinds (discrete trace, LM iteration count, pose);
# the doc-binding test; the backward sweep of the slot teams
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04u
mkdir -p $O
timeout 300 python -m pytest tests/test_integration_doc.py -m gpu -q 2>&1 | tail -2
timeout 1500 python scripts/dev/sweep.py 3000 > $O/r04_sweep_3000.txt 2>&1
tail -5 $O/r04_sweep_3000.txt
