#!/bin/bash
# round 4: team sizes once more, after the instruction trims (the serial section shrank: does a wider team pay now?)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04sizes
run() { timeout 120 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-training --batch 0 --no-exact --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('value %.0f ms %.4f members %s refine stage %.4f' % (d['value'], d['ms_per_step'], d['refine'].get('workgroups'), d['phase_ms']['refine']))"; }
for g in 8 12 19 8 19; do echo "== ESAC_REFINE_TEAM=$g"; ESAC_REFINE_TEAM=$g run; done 2>&1 | tee gpurun_out/r04sizes/sizes.txt
