#!/usr/bin/env python
"""Generates tests/golden/ref_*.npz from the REFERENCE ITSELF: esac_forward / esac_backward compiled from
/root/reference/code/esac/*.cpp|h (oracle/_ref, built by `make -C oracle ref`; OpenCV/ATen are stand-in shims whose
numerics forward to the oracle's restatements) on small seeded cases, single thread.

Besides inputs and the reference's outputs each fixture records the integer stream the reference's mt19937 handed
out (irand(lo,hi) draws, thread_rand.cpp:68-71), so the oracle -- whose callback RNG replays it -- can be checked
against these vectors where /root/reference does not exist (the GPU box, CI).  tests/test_ref_golden.py is that check.
Needs /root/reference; re-run only after an intended change:   python tests/golden/make_ref_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from esac_amd import synthetic as S  # noqa: E402
from oracle import esac_oracle as O  # noqa: E402
from oracle import ref_binding as R  # noqa: E402

SEED = 1305
CASES = {
    "ref_fwd_small": dict(kind="forward", frame=dict(k=20, H=24, W=32, sub=20), N=48),
    "ref_fwd_2experts": dict(kind="forward", frame=dict(k=21, H=30, W=40, sub=16, E=2, true_expert=1), N=64, mode="gating"),
    "ref_bwd_small": dict(kind="backward", frame=dict(k=22, H=24, W=32, sub=20), N=32, loss=dict()),
    "ref_bwd_clamped": dict(kind="backward", frame=dict(k=23, H=24, W=32, sub=20, shift=(3, -2)), N=32,
                            loss=dict(w_rot=2.0, w_trans=50.0, loss_cut=0.5)),
    "ref_bwd_3experts": dict(kind="backward", frame=dict(k=24, H=24, W=32, sub=20, E=3, true_expert=2), N=48, mode="gating",
                             loss=dict(), params=dict(inlier_alpha=10.0)),
    "ref_fwd_params": dict(kind="forward", frame=dict(k=25, H=27, W=35, sub=18, shift=(-4, 5)), N=40,
                           params=dict(inlier_thresh=6.0, inlier_alpha=50.0, inlier_beta=0.8, max_reproj=60.0)),
}


def record_stream(seed, run):
    """Runs `run(irand)` with the oracle on the reference's mt19937 stream and returns the draws it consumed."""
    draws = []
    replay = R.replay_irand(seed)

    def irand(lo, hi):
        v = replay(lo, hi)
        draws.append((lo, hi, v))
        return v
    out = run(irand)
    return out, np.array(draws, np.int32)


def main():
    assert R.build() is not None, "oracle/_ref could not be built (is /root/reference mounted?)"
    for name, c in CASES.items():
        f = S.make_frame(**c["frame"])
        ha = S.gating_assignment(f, c["N"], mode=c.get("mode", "single"))
        kw = dict(shift_x=f["shift"][0], shift_y=f["shift"][1], focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"])
        kw.update(c.get("params", {}))
        params = dict(inlier_thresh=10.0, inlier_alpha=100.0, inlier_beta=0.5, max_reproj=100.0)
        params.update(c.get("params", {}))
        out = dict(coords=f["coords"], assign=ha, focal=np.float32(f["focal"]), ppx=np.float32(f["ppx"]), ppy=np.float32(f["ppy"]),
                   sub=np.int32(f["sub"]), shift=np.array(f["shift"], np.int32), kind=c["kind"],
                   **{k: np.float32(v) for k, v in params.items()})
        if c["kind"] == "forward":
            expert, pose = R.esac_forward(f["coords"], ha, seed=SEED, **kw)          # the reference's esac_forward
            staged = R.forward(f["coords"], ha, seed=SEED, **kw)                     # its stages, same stream
            ora, draws = record_stream(SEED, lambda ir: O.forward(f["coords"], ha, irand=ir, **kw))
            assert expert == ora["expert"] and np.array_equal(pose, ora["pose"])
            out.update(ref_expert=np.int32(expert), ref_pose=pose, ref_sample_xy=staged["sample_xy"], ref_hyps=staged["hyps"],
                       ref_scores=staged["scores"], ref_winner=np.int32(staged["winner"]), ref_refined=staged["refined"],
                       ref_inlier_map=staged["inlier_map"])
        else:
            gt = f["gt_pose"].astype(np.float32)
            gt[:3, 3] += np.float32(0.03)
            g_ref = np.zeros_like(f["coords"])
            loss = R.esac_backward(f["coords"], g_ref, ha, gt, seed=SEED, **kw, **c["loss"])  # the reference's esac_backward
            g_ora = np.zeros_like(f["coords"])
            ora, draws = record_stream(SEED, lambda ir: O.backward(f["coords"], g_ora, ha, gt, irand=ir, **kw, **c["loss"]))
            assert np.array_equal(g_ora, g_ref)
            lk = dict(w_rot=1.0, w_trans=100.0, loss_cut=100.0)
            lk.update(c["loss"])
            out.update(gt_pose=gt, ref_loss=np.float64(loss), ref_gradients=g_ref, w_rot=np.float32(lk["w_rot"]),
                       w_trans=np.float32(lk["w_trans"]), loss_cut=np.float32(lk["loss_cut"]))
        out["draws"] = draws
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path) // 1024, "KiB,", len(draws), "draws")


if __name__ == "__main__":
    main()
