"""Forward calls of one BASELINE workload for a rocprofv3 kernel trace (scripts/dev/timeline.sh).
usage: fwd_loop.py <cfg2|cfg3|cfg4|cfg5a> [calls]"""
import os
import sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from esac_amd import api, synthetic as S
cfg = {"cfg2": (1, 256, "single"), "cfg3": (10, 1024, "gating"), "cfg4": (12, 4096, "gating"), "cfg5a": (50, 16384, "dirichlet")}[sys.argv[1]]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
E, N, mode = cfg
eng = api.engine(0)
frames = [S.make_frame(k, E=E) for k in range(8)]
sc = [torch.from_numpy(f["coords"]).cuda() for f in frames]
ha = [torch.from_numpy(S.gating_assignment(f, N, mode=mode if E > 1 else "single")).cuda() for f in frames]
p = eng.make_params(E, 60, 80, N, seed=1320, call=0, exact_scores="auto")
for i in range(calls):
    p.call = i
    eng.forward_device(sc[i % 8], ha[i % 8], p)
print(sys.argv[1], "calls", calls, "spec", eng.spec_info())
