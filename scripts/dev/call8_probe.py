import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from esac_amd import api, synthetic as S
eng = api.engine(0)
frames = [S.make_frame(k) for k in range(16)]
assigns = [S.gating_assignment(f, 256, mode="single") for f in frames]
d_coords = [torch.from_numpy(f["coords"]).cuda() for f in frames]
d_assign = [torch.from_numpy(a).cuda() for a in assigns]
scores = torch.empty(256, dtype=torch.float64, device="cuda")
params = eng.make_params(1, 60, 80, 256, seed=1320, call=0, focal=frames[0]["focal"], ppx=frames[0]["ppx"], ppy=frames[0]["ppy"], sub_sampling=8)
for i in range(26):
    params.call = i
    ts = []
    for rep in range(5):
        t0 = time.perf_counter_ns()
        r = eng.forward_device(d_coords[i % 16], d_assign[i % 16], params, scores_out=scores)
        ts.append((time.perf_counter_ns() - t0) * 1e-3)
    info = eng.refine_info()
    tries = eng.read(api.BUF_TRIES)
    print("call %2d: %6.1f us (min of 5) steps %d lm %2d exchanges %3d contenders %d max tries %d inliers %d" % (i, min(ts[1:]), int(r[api.RES_REF_STEPS]), int(r[api.RES_LM_ITERS]), info["exchanges"], int(r[api.RES_CONTENDERS]), int(tries.max()), int(r[api.RES_INLIERS])))
