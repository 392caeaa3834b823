#!/usr/bin/env python
"""Calibration run of the fp32 sampling screen (host build of esac_amd/csrc/p3p_screen.hpp vs the fp64 route):
python scripts/dev/p3p_screen_probe.py [tries per map, default 2e6]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from esac_amd import synthetic as S  # noqa: E402

from tests.native import build as nb  # noqa: E402


def build():
    return C.CDLL(nb.build_screen_probe())


def run(lib, name, coords, f, n, seed=1, mode=None):
    mode = MODE if mode is None else mode
    margins = np.array([0.5, 1, 2, 5, 10, 20, 40, 80], np.float32)
    out = np.zeros(40)
    c = np.ascontiguousarray(coords, np.float32)
    _, H, W = c.shape
    lib.probe_screen(c.ctypes.data_as(C.c_void_p), H, W, f["sub"], f["shift"][0], f["shift"][1], C.c_float(f["focal"]), C.c_float(f["ppx"]),
                     C.c_float(f["ppy"]), C.c_float(10.0), C.c_uint64(seed), C.c_longlong(int(n)), margins.ctypes.data_as(C.c_void_p), 8, mode,
                     out.ctypes.data_as(C.c_void_p))
    t, acc = out[0], out[1]
    print("%-28s tries %.2e accepted %8d (%.2e) delicate %.3f%%  max screen err of an accepted try %.3f px  max |e32-e64| %.3f" % (
        name, t, acc, acc / t, 100 * out[2] / t, out[3], out[20]))
    print("      bail-outs (quartic triggers 1-5, other): " + " ".join("%.3f%%" % (100 * v / t) for v in out[23:29]))
    print("      margin      : " + " ".join("%9g" % m for m in margins))
    print("      maybe frac  : " + " ".join("%9.5f" % (v / t) for v in out[4:12]))
    print("      false reject: " + " ".join("%9d" % v for v in out[12:20]))
    return out


if __name__ == "__main__":
    n = float(sys.argv[1]) if len(sys.argv) > 1 else 2e6
    MODE = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
    lib = build()
    if len(sys.argv) > 2 and sys.argv[2] == "debug":
        f = S.make_frame(0, E=2, true_expert=0)
        run(lib, "frame 0 true expert", f["coords"][0], f, n, seed=0, mode=2)
        sys.exit(0)
    for k in range(3):
        f = S.make_frame(k, E=2, true_expert=0)
        run(lib, "frame %d true expert" % k, f["coords"][0], f, n, seed=k)
        run(lib, "frame %d garbage expert" % k, f["coords"][1], f, n, seed=100 + k)
    f = S.make_frame(7, noise=0.0, outlier_frac=0.0)
    run(lib, "noise-free map", f["coords"][0], f, n, seed=9)
    f = S.make_frame(8, H=24, W=32, sub=20)
    run(lib, "24x32 sub 20", f["coords"][0], f, n, seed=10)
    f = S.make_frame(9, E=2, true_expert=0)
    off = np.array([1200.0, -800.0, 950.0], np.float32)
    run(lib, "world offset 1e3 true", f["coords"][0] + off[:, None, None], f, n, seed=11)
    run(lib, "world offset 1e3 garbage", f["coords"][1] + off[:, None, None], f, n, seed=12)
