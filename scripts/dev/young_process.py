"""Do the kernels of the headline call get faster over the first calls of a process?  The SAME frame and RNG key every call (the
same work), 80 blocking calls; run under rocprofv3 --kernel-trace (scripts/dev/young_process.sh) the per-call durations are read
from the trace; stand-alone it prints the call durations as the host sees them."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from esac_amd import api, synthetic as S
eng = api.engine(0)
f = S.make_frame(3)
sc = torch.from_numpy(f["coords"]).cuda(); ha = torch.from_numpy(S.gating_assignment(f, 256)).cuda()
scores = torch.empty(256, dtype=torch.float64, device="cuda")
p = eng.make_params(1, 60, 80, 256, seed=1320, call=7, exact_scores="auto")
ts = []
for i in range(80):
    t0 = time.perf_counter_ns()
    eng.forward_device(sc, ha, p, scores_out=scores)
    ts.append((time.perf_counter_ns() - t0) * 1e-3)
for i in range(0, 80, 10):
    print("calls %2d..%2d (us):" % (i, i + 9), " ".join("%6.1f" % t for t in ts[i:i + 10]))
