#!/usr/bin/env python
"""Where the headline call's time goes beyond its kernels, and whether several workgroups should share the refinement of a
60x80 grid: (1) blocking esac_hip_forward at cfg2 through the Engine wrapper and through a bare ctypes call with prebuilt
arguments; (2) the same on a tiny problem (fixed cost of 4 launches + host path).
(Round 3 also timed G = 2..8 cooperating workgroups on the 60x80 grid through a debug knob: 155-168 us against 148 us for
one workgroup -- a barrier round costs more than the work it splits)"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from esac_amd import api, synthetic as S  # noqa: E402

eng = api.engine(0)
frames = [S.make_frame(k) for k in range(16)]
d_sc = [torch.from_numpy(f["coords"]).cuda() for f in frames]
d_ha = [torch.from_numpy(S.gating_assignment(f, 256)).cuda() for f in frames]
p = eng.make_params(1, 60, 80, 256, seed=1320, call=0)
scores = torch.empty(256, dtype=torch.float64, device="cuda")


def run_engine(n, first=0):
    for i in range(n):
        p.call = first + i
        eng.forward_device(d_sc[i % 16], d_ha[i % 16], p, scores_out=scores)


def timed(fn, n, *a):
    fn(40, *a)
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn(n, *a)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


print("cfg2 blocking call through Engine.forward_device: %.1f us" % timed(run_engine, 400, 40))
lib, host = eng.lib, np.zeros(32)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ptrs = [(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr())) for a, b in zip(d_sc, d_ha)]
sp, hp, pp = C.c_void_p(scores.data_ptr()), C.c_void_p(host.ctypes.data), C.byref(p)


def run_raw(n, first=0):
    for i in range(n):
        p.call = first + i
        a, b = ptrs[i % 16]
        lib.esac_hip_forward(eng.ctx, a, b, pp, stream, sp, None, hp)


print("cfg2 blocking call, bare ctypes with prebuilt arguments: %.1f us" % timed(run_raw, 400, 40))
st = {k: 0.0 for k in ("sample", "score", "select_rescore", "refine")}
for k in range(16):
    p.call = 40 + k
    for n, v in eng.time_stages(d_sc[k], d_ha[k], p, 12).items():
        st[n] += v / 16 * 1e3
print("stages (us):", {k: round(v, 1) for k, v in st.items()}, "sum %.1f" % sum(st.values()))
ft = S.make_frame(0, H=12, W=16, sub=40)
tsc, tha = torch.from_numpy(ft["coords"]).cuda(), torch.from_numpy(S.gating_assignment(ft, 8)).cuda()
tp = eng.make_params(1, 12, 16, 8, sub_sampling=40, max_ref_steps=0)


def run_tiny(n):
    for i in range(n):
        tp.call = i
        eng.forward_device(tsc, tha, tp)


print("tiny problem (8 hypotheses, 12x16 grid, no refinement step) blocking call: %.1f us" % timed(run_tiny, 1000))
