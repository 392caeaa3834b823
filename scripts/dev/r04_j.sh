#!/bin/bash
# score kernels: 4 transcendentals per cell (rcp, sqrt, exp2, rcp) against 3 (rsq, exp2, rcp) -- per-kernel rocprofv3 averages, same box
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j
mkdir -p $O
bash scripts/dev/kvariants.sh "--config cfg5b --steps 6 --warmup 2 --no-exact" "k_score" "base=" "rsq=-DESAC_SCORE_RSQ" > $O/score_rsq_cfg5b.txt 2>&1
cat $O/score_rsq_cfg5b.txt
bash scripts/dev/kvariants.sh "--config cfg5a --steps 40 --warmup 5 --no-exact" "k_score" "base=" "rsq=-DESAC_SCORE_RSQ" > $O/score_rsq_cfg5a.txt 2>&1
cat $O/score_rsq_cfg5a.txt
python scripts/dev/latency_probe.py 2>&1 | tail -12
