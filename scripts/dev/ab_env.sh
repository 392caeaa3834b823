#!/bin/bash
# same-box A/B of environment settings on one bench workload: scripts/dev/ab_env.sh "<bench args>" VAR=val [VAR=val ...]  (each
# setting once per round, ROUNDS rounds interleaved; the line's value, ms_per_step and the live refine stage)
cd ${GRAFT_REPO_ROOT:-.}
ARGS=$1; shift
for r in $(seq 1 ${ROUNDS:-2}); do
  for v in "$@"; do
    env $v python bench.py $ARGS --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %s %.4f ms  %.3f M hyp/s  stages %s' % ('$v', d['config']['name'], d['ms_per_step'], d['value']/1e6, [round(k['avg_us'],1) for k in d.get('kernels', [])]))"
  done
done
