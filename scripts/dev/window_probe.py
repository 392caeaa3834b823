#!/usr/bin/env python
"""How representative is the driver's short bench window (--warmup 5 --steps 20) of the steady state?  The work of one
esac.forward depends on the refinement path of its winner, i.e. on (frame, RNG key): for the bench's cycled frames this
prints, per candidate seed, the mean refinement steps / LM iterations / measured ms over calls 5..24 and over 400 calls.
python scripts/dev/window_probe.py [first_seed] [n_seeds]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from esac_amd import api, synthetic as S  # noqa: E402

dev = torch.device("cuda", 0)
frames = [S.make_frame(k, E=1, H=60, W=80, sub=8) for k in range(16)]
assigns = [S.gating_assignment(f, 256, mode="single") for f in frames]
eng = api.engine(0)
kw = dict(focal=frames[0]["focal"], ppx=frames[0]["ppx"], ppy=frames[0]["ppy"], sub_sampling=8)
d_a = [torch.from_numpy(a).to(dev) for a in assigns]
d_c = [torch.from_numpy(f["coords"]).to(dev) for f in frames]
scores = torch.empty(256, dtype=torch.float64, device=dev)
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1305
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 24
for seed in range(s0, s0 + ns):
    params = eng.make_params(1, 60, 80, 256, seed=seed, call=0, **kw)
    steps, iters, ms = [], [], []
    for i in range(445):
        params.call = i
        t0 = time.perf_counter()
        r = eng.forward_device(d_c[i % 16], d_a[i % 16], params, scores_out=scores)
        ms.append((time.perf_counter() - t0) * 1e3)
        steps.append(r[api.RES_REF_STEPS])
        iters.append(r[api.RES_LM_ITERS])
    steps, iters, ms = np.array(steps), np.array(iters), np.array(ms)
    print("seed %d  window 5..24: steps %.2f iters %.1f ms %.4f | calls 45..444: steps %.2f iters %.1f ms %.4f | ratio %.3f" % (
        seed, steps[5:25].mean(), iters[5:25].mean(), ms[5:25].mean(), steps[45:].mean(), iters[45:].mean(), ms[45:].mean(),
        ms[5:25].mean() / ms[45:].mean()))
