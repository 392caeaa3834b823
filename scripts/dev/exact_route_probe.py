#!/usr/bin/env python
"""Device exact route (pose_math.hpp compiled for the HOST: same algorithm, glibc's libm) against the oracle on the
adversarial maps of the GPU test tests/test_gpu_semantics.py::test_screened_sampling_on_adversarial_geometry: where the two
disagree on the accepted try, the device's ALGORITHM (triad + Newton alignment instead of Horn's eigenvector, ...) is what
differs, not the GPU's libm.  python scripts/dev/exact_route_probe.py [kind ...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts", "dev"))
import screen_adversarial as A  # noqa: E402
from oracle import esac_oracle as O  # noqa: E402
from tests.native import build as nb  # noqa: E402

KINDS = {"planar": ["plane warped 3x0.33 tilted", "plane warped 2x0.5"],
         "degenerate": ["points within 1 mm of a line", "plane warped, x and y quantised"],
         "curved": ["sphere warped 2x0.5", "room, rows swapped pairwise"]}

if __name__ == "__main__":
    lib = C.CDLL(nb.build_screen_probe())
    maps = A.adversarial_maps()
    N, max_tries = int(os.environ.get("N", 12288)), 5000
    for kind in (sys.argv[1:] or list(KINDS)):
        for e, name in enumerate(KINDS[kind], start=1):
            gh = np.arange(e, N, 3).astype(np.int32)  # the hypotheses the GPU test puts on this expert
            c = np.ascontiguousarray(maps[name][None])
            o = O.forward(c, np.zeros(len(gh), np.int64), seed=77, call=3, max_tries=max_tries, hyp_index=gh)
            mine = np.zeros(len(gh), np.int32)
            lib.probe_first_accept(c.ctypes.data_as(C.c_void_p), 60, 80, 8, C.c_float(525.0), C.c_float(320.0), C.c_float(240.0), C.c_float(10.0),
                                   C.c_uint64(77), C.c_uint64(3), gh.ctypes.data_as(C.c_void_p), len(gh), max_tries, mine.ctypes.data_as(C.c_void_p))
            bad = np.nonzero(mine != o["tries"])[0]
            tries = np.where(o["tries"] < 0, max_tries, o["tries"] + 1).sum()
            print("%-12s %-34s hypotheses %d tries %.2e  device-math-on-host != oracle: %d %s" % (
                kind, name, len(gh), tries, len(bad), [(int(gh[i]), int(mine[i]), int(o["tries"][i])) for i in bad[:6]]))
