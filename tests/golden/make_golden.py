#!/usr/bin/env python
"""Generates tests/golden/*.npz: inputs + every stage output of the CPU oracle on small seeded cases.

The reference ships no golden vectors and cannot be built here (OpenCV), so these fixtures are produced by
the oracle (oracle/esac_oracle.c) -- they pin the oracle against regressions and travel to the GPU box,
where /root/reference and a second toolchain do not exist.  Re-run after an INTENDED oracle change:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from esac_amd import synthetic as S  # noqa: E402
from oracle import esac_oracle as O  # noqa: E402

CASES = {
    # name: (frame kwargs, N, assignment mode, forward kwargs)
    "cfg1_1expert_64hyp": (dict(k=0), 64, "single", dict(call=0)),
    "cfg2_1expert_256hyp": (dict(k=1), 256, "single", dict(call=1)),
    "gating_3experts_128hyp": (dict(k=2, E=3, true_expert=1), 128, "gating", dict(call=2)),
    "small_grid_shifted": (dict(k=3, H=24, W=32, sub=20, shift=(3, -2)), 32, "single", dict(call=3, shift_x=3, shift_y=-2)),
}
KEEP = ("pose", "sample_xy", "tries", "hyps", "scores", "refined", "inlier_counts", "inlier_map")


def main():
    for name, (fkw, N, mode, kw) in CASES.items():
        f = S.make_frame(**fkw)
        ha = S.gating_assignment(f, N, mode=mode)
        o = O.forward(f["coords"], ha, focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"],
                      seed=1305, num_threads=1, **kw)
        out = {k: o[k] for k in KEEP}
        out.update(coords=f["coords"], assign=ha, winner=np.int32(o["winner"]), expert=np.int32(o["expert"]),
                   ref_steps=np.int32(o["ref_steps"]), entropy=np.float64(o["entropy"]),
                   focal=np.float32(f["focal"]), ppx=np.float32(f["ppx"]), ppy=np.float32(f["ppy"]),
                   sub=np.int32(f["sub"]), shift=np.array(f["shift"], np.int32), seed=np.uint64(1305),
                   call=np.uint64(kw["call"]), gt_pose=f["gt_pose"])
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path) // 1024, "KiB", "winner", o["winner"], "steps", o["ref_steps"])


if __name__ == "__main__":
    main()
