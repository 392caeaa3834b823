#!/bin/bash
# slot teams of the training path: the backward tests, then backward time with / without teams on one box
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04t
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backward.py -m gpu -q > $O/pytest.log 2>&1
grep -E "passed|failed|Error|assert" $O/pytest.log | head -20
for rep in 1 2; do
for t in 0 1; do
  ESAC_SLOT_TEAMS=$t timeout 300 python bench.py --no-cpu-baseline --no-exact --batch 0 > $O/bench_$t.json 2> $O/bench_$t.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_$t.json").read().strip().splitlines()[-1])
    print("slot teams $t: backward %.4f ms, refined per call %.1f" % (d["training"]["ms_per_call"], d["training"]["refined_hypotheses_per_call"]))
except Exception as e:
    print("$t FAILED", e); print(open("$O/bench_$t.err").read()[-2000:])
PY
done
done
