"""The driver's window (calls 5..24 of a fresh process run ~4-5 us slower than a long run, on the GPU side): does a burst of TINY launches
in front of the first call remove it?  usage: python scripts/dev/prewarm_probe.py <tiny launches> [kind: torch | lib]"""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from esac_amd import api, synthetic as S

n_tiny = int(sys.argv[1]) if len(sys.argv) > 1 else 0
kind = sys.argv[2] if len(sys.argv) > 2 else "torch"
eng = api.engine(0)
frames = [S.make_frame(k) for k in range(16)]
assigns = [S.gating_assignment(f, 256, mode="single") for f in frames]
d_c = [torch.from_numpy(f["coords"]).cuda() for f in frames]
d_a = [torch.from_numpy(a).cuda() for a in assigns]
scores = torch.empty(256, dtype=torch.float64, device="cuda")
p = eng.make_params(1, 60, 80, 256, seed=1320, call=0, exact_scores="auto")
torch.cuda.synchronize()
if n_tiny:
    if kind == "torch":
        x = torch.zeros(64, device="cuda")
        for _ in range(n_tiny):
            x.add_(1.0)
    elif kind == "fwd":  # whole forward calls on ONE other frame (the kernels of the timed calls, on the CUs they use)
        f1 = S.make_frame(99)
        a1 = torch.from_numpy(S.gating_assignment(f1, 256, mode="single")).cuda()
        c1 = torch.from_numpy(f1["coords"]).cuda()
        p1 = eng.make_params(1, 60, 80, 256, seed=7, call=0, exact_scores="auto")
        for j in range(n_tiny):
            p1.call = j
            eng.forward_device(c1, a1, p1)
    else:  # the library's own smallest launches: the sampling stage of a 1-hypothesis problem
        f1 = S.make_frame(99)
        a1 = torch.from_numpy(S.gating_assignment(f1, 4, mode="single")).cuda()
        c1 = torch.from_numpy(f1["coords"]).cuda()
        p1 = eng.make_params(1, 60, 80, 4, seed=1, call=0)
        for _ in range(n_tiny):
            eng.sample(c1, a1, p1)
    torch.cuda.synchronize()


def step(i):
    p.call = i
    return eng.forward_device(d_c[i % 16], d_a[i % 16], p, scores_out=scores)


for i in range(5):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(5, 25):
    step(i)
torch.cuda.synchronize()
t1 = time.perf_counter()
for i in range(25, 65):
    step(i)
torch.cuda.synchronize()
t2 = time.perf_counter()
for i in range(65, 465):
    step(i)
torch.cuda.synchronize()
t3 = time.perf_counter()
print("tiny launches %4d (%s): calls 5..24 %.4f ms | 25..64 %.4f ms | 65..464 %.4f ms" % (n_tiny, kind, (t1 - t0) / 20 * 1e3, (t2 - t1) / 40 * 1e3, (t3 - t2) / 400 * 1e3))
