"""round 5: duration of each of the first 80 blocking calls of a fresh process (the driver times steps 6..25 of such a process)."""
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
params = eng.make_params(1, 60, 80, 256, seed=1320, call=0, focal=frames[0]["focal"], ppx=frames[0]["ppx"], ppy=frames[0]["ppy"], sub_sampling=8, exact_scores=False if os.environ.get("ESAC_FC_FAST") else "auto")
torch.cuda.synchronize()
ts, steps, lm = [], [], []
for i in range(80):
    params.call = i
    t0 = time.perf_counter_ns()
    r = eng.forward_device(d_coords[i % 16], d_assign[i % 16], params, scores_out=scores)
    ts.append((time.perf_counter_ns() - t0) * 1e-3)
    h = eng.host_turn()
    steps.append((int(r[api.RES_REF_STEPS]), int(r[api.RES_LM_ITERS]), round(h["sample_launched"], 1), round(h["record_landed"] - h["refine_launched"], 1)))
for i in range(0, 80, 8):
    print("calls %2d..%2d us:" % (i, i + 7), " ".join("%6.1f" % t for t in ts[i:i + 8]), "| (steps, lm, first launch us, wait us):", steps[i:i + 8][:3])
print("mean of calls 5..24: %.1f us; of calls 40..79: %.1f us" % (np.mean(ts[5:25]), np.mean(ts[40:80])))
print("LM iterations: calls 5..24 mean %.2f, calls 40..79 mean %.2f" % (np.mean([s[1] for s in steps[5:25]]), np.mean([s[1] for s in steps[40:80]])))
