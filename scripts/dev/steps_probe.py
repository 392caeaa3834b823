#!/usr/bin/env python
"""Per-call latency of esac.forward from a cold process: ONE frame and ONE RNG key repeated (identical work every call), so
what changes over the calls is the device (clock ramp, caches), not the workload.  python scripts/dev/steps_probe.py"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from esac_amd import api, synthetic as S  # noqa: E402

dev = torch.device("cuda", 0)
f = S.make_frame(0, E=1, H=60, W=80, sub=8)
a = S.gating_assignment(f, 256, mode="single")
eng = api.engine(0)
kw = dict(focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=8)
d_a = torch.from_numpy(a).to(dev)
d_c = torch.from_numpy(f["coords"]).to(dev)
scores = torch.empty(256, dtype=torch.float64, device=dev)
params = eng.make_params(1, 60, 80, 256, seed=1305, call=7, **kw)
torch.cuda.synchronize()


def burst(n):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        eng.forward_device(d_c, d_a, params, scores_out=scores)
        ts.append((time.perf_counter() - t0) * 1e3)
    return np.array(ts)


ts = burst(400)
print("cold process, identical calls (ms):", " ".join("%.3f" % t for t in ts[:30]))
for lo, hi in ((1, 6), (6, 26), (26, 50), (50, 100), (100, 200), (200, 400)):
    print("  calls %3d..%3d  mean %.4f  min %.4f" % (lo, hi, ts[lo:hi].mean(), ts[lo:hi].min()))
for idle in (0.01, 0.1, 1.0):
    time.sleep(idle)
    t2 = burst(60)
    print("after %.2f s idle: first 8:" % idle, " ".join("%.3f" % t for t in t2[:8]), " mean 0..20 %.4f  mean 40..60 %.4f" % (t2[:20].mean(), t2[40:].mean()))
