#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -12 $O/pytest.log
for rep in 1 2 3; do
for fold in 1 0; do
  ESAC_FOLD_SELECT=$fold timeout 300 python bench.py --no-cpu-baseline --no-extras > $O/bench_fold$fold.json 2> $O/bench_fold$fold.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_fold$fold.json").read().strip().splitlines()[-1])
    print("fold $fold: ms %.4f value %.0f seed1305 %s" % (d["ms_per_step"], d["value"], d.get("value_seed1305")), {k["stage"]: round(k["avg_us"],1) for k in d.get("kernels",[])})
except Exception as e:
    print("fold $fold FAILED", e); print(open("$O/bench_fold$fold.err").read()[-2000:])
PY
done
done
python - <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, '.')
from esac_amd import api, synthetic as S
eng = api.engine(0)
nc = []
for k in range(64):
    f = S.make_frame(k); ha = S.gating_assignment(f, 256)
    p = eng.make_params(1, 60, 80, 256, seed=1320, call=k)
    r = eng.forward_device(torch.from_numpy(f['coords']).cuda(), torch.from_numpy(ha).cuda(), p)
    nc.append(int(r[api.RES_CONTENDERS]))
print('contenders per frame: mean %.2f max %d hist %s' % (np.mean(nc), max(nc), np.bincount(nc)[:12]))
PY
