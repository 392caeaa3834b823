"""round 5: the team refinement stage reads ~67 or ~71 us from process to process on one box.  Does the state follow what was
launched before (dispatcher position), i.e. does it flip INSIDE a process when other launches are interleaved?"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from esac_amd import api, synthetic as S
eng = api.engine(0)
f = S.make_frame(3)
ha = S.gating_assignment(f, 256, mode="single")
sc, hat = torch.from_numpy(f["coords"]).cuda(), torch.from_numpy(ha).cuda()
p = eng.make_params(1, 60, 80, 256, seed=1320, call=5, exact_scores="auto")
x = torch.zeros(1 << 20, device="cuda")
for k in range(3):
    eng.forward_device(sc, hat, p)
for trial in range(24):
    n_dummy = trial % 12
    for _ in range(n_dummy):
        x.add_(1.0)  # 4096-workgroup elementwise launches: move whatever round-robin state the dispatcher keeps
    torch.cuda.synchronize()
    st = eng.time_stages(sc, hat, p, reps=20)
    info = eng.refine_info()
    print("dummy launches %2d: refine %.2f us  sample %.2f  score %.2f  census %s" % (n_dummy, st["refine"] * 1e3, st["sample"] * 1e3, st["score"] * 1e3, info["xcd_census"]))
