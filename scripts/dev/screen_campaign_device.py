#!/usr/bin/env python
"""The sampling screen against the fp64 route ON THE DEVICE (tests/native/screen_campaign.hip): the adversarial map
families of screen_adversarial.py, N tries per map (default 2e9), the kernels' arithmetic instead of the host's.
python scripts/dev/screen_campaign_device.py [tries per map] [adversarial|quantised|both]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tests.native import build as nb  # noqa: E402
from screen_adversarial import adversarial_maps, quantised_maps  # noqa: E402


def run(lib, coords, n, seed, sub=8, focal=525.0, ppx=320.0, ppy=240.0, tau=10.0, blocks=8192, per_thread=256):
    c = np.ascontiguousarray(coords, np.float32)
    _, H, W = c.shape
    launches = max(1, int(round(n / (blocks * 64.0 * per_thread))))
    out = np.zeros(9)
    rc = lib.screen_campaign(c.ctypes.data_as(C.c_void_p), H, W, sub, C.c_float(focal), C.c_float(ppx), C.c_float(ppy), C.c_float(tau),
                             C.c_uint64(seed), blocks, per_thread, launches, out.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    return out


if __name__ == "__main__":
    n = float(sys.argv[1]) if len(sys.argv) > 1 else 2e9
    which = sys.argv[2] if len(sys.argv) > 2 else "both"
    lib = C.CDLL(nb.build_screen_campaign())
    fams = []
    if which in ("adversarial", "both"):
        fams.append(("adversarial", adversarial_maps(), 500))
    if which in ("quantised", "both"):
        fams.append(("quantised", quantised_maps(), 900))
    tot = np.zeros(9)
    t0 = time.time()
    for fname, fam, seed0 in fams:
        for k, (name, coords) in enumerate(fam.items()):
            if name == "points on a line":
                continue  # (the known divergence of the exact route itself on exactly collinear maps, DESIGN.md: not a screen question)
            out = run(lib, coords, n, seed0 + k)
            tot[:8] += out[:8]
            tot[8] = max(tot[8], out[8])
            print("%-40s tries %.2e  fp64-accepted %11d (%.2e)  to the fp64 decision %.4f%%  delicate %.3f%%  false rejects @0.5/1/2/3 px: %d %d %d %d  max screen err of accepted %.3f"
                  % (name, out[0], out[1], out[1] / out[0], 100 * out[3] / out[0], 100 * out[2] / out[0], out[4], out[5], out[6], out[7], out[8]), flush=True)
    print("TOTAL on the device: %.3e tries, %.3e accepted by the fp64 route, false rejects at 0.5 / 1 / 2 / 3 px: %d %d %d %d (the kernels use 3 px), "
          "largest screen error of an accepted try %.3f px (tau = 10), %.0f s" % (tot[0], tot[1], tot[4], tot[5], tot[6], tot[7], tot[8], time.time() - t0))
