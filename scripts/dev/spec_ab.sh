#!/bin/bash
# same-box A/B of the speculative forward's knobs: bench value / ms_per_step of cfg3 and cfg4 per environment setting
cd ${GRAFT_REPO_ROOT:-.}
run() {
  for c in cfg3 cfg4; do
    env "$@" python bench.py --config $c --no-cpu-baseline --no-extras --steps 300 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %s %.4f ms  %.3f M hyp/s' % ('$*', d['config']['name'], d['ms_per_step'], d['value']/1e6))"
  done
}
for v in "$@"; do run $v; done
