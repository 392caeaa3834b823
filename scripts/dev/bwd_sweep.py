"""Soak of the training path (slot teams from the second call on) against the oracle: expected loss, slot list, refined poses,
gradient tensor over many frames.  python scripts/dev/bwd_sweep.py [frames, default 200]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from esac_amd import api, synthetic as S  # noqa: E402
from oracle import esac_oracle as O  # noqa: E402

eng = api.Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
worst = {"loss": 0.0, "pose": 0.0, "grad": 0.0, "grad_sampled": 0.0}
teams = 0
t0 = time.time()
for k in range(n):
    E = 1 if k % 3 else 3
    f = S.make_frame(2000 + k, E=E, true_expert=k % E)
    N = (64, 128, 256)[k % 3]
    ha = S.gating_assignment(f, N, mode="gating" if E > 1 else "single")
    gt = np.array(f["gt_pose"], np.float32)
    gt[:3, 3] += np.float32(0.02 * (k % 5))
    alpha = (100.0, 30.0)[k % 2]
    sc, hat = torch.from_numpy(f["coords"]).cuda(), torch.from_numpy(ha).cuda()
    p = eng.make_params(E, 60, 80, N, focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"], inlier_alpha=alpha, seed=55, call=k)
    g = torch.zeros_like(sc)
    out = eng.backward_device(sc, g, hat, gt, 1.0, 100.0, 100.0, p)
    teams += eng.bwd_team_info()["teams"]
    g_ref = np.zeros_like(f["coords"])
    ref = O.backward(f["coords"], g_ref, ha, gt, w_rot=1.0, w_trans=100.0, loss_cut=100.0, focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"],
                     sub_sampling=f["sub"], inlier_alpha=alpha, seed=55, call=k)
    sel_ref = np.nonzero(ref["probs"] >= 1e-3)[0]
    edge = (np.abs(ref["probs"] - 1e-3) < 1e-12).any()
    ok = edge or (int(out[1]) == len(sel_ref) and np.array_equal(eng.read(api.BUF_BWD_SLOTS)[:len(sel_ref)], sel_ref))
    dl = abs(out[0] - ref["loss"]) / max(1.0, abs(ref["loss"]))
    dp = float(np.abs(eng.read(api.BUF_BWD_REF_HYPS) - ref["ref_hyps"]).max())
    scale = max(float(np.abs(g_ref).max()), 1e-30)
    sampled = np.zeros((E, 60, 80), bool)
    for h in sel_ref:
        for x, y in ref["sample_xy"][h]:
            sampled[:, y, x] = True
    diff = np.abs(g.cpu().numpy() - g_ref)
    dg = float(diff[:, :, ~sampled.any(0)].max()) / scale
    dgs = float(diff[:, :, sampled.any(0)].max()) / scale if sampled.any() else 0.0
    for key, v in (("loss", dl), ("pose", dp), ("grad", dg), ("grad_sampled", dgs)):
        worst[key] = max(worst[key], v)
    if not (ok and dl <= 1e-7 and dp <= 1e-6 and dg <= 5e-7 and dgs <= 1e-3):
        bad += 1
        print("MISMATCH frame", k, "slots", int(out[1]), len(sel_ref), "loss", dl, "pose", dp, "grad", dg, dgs)
print("backward calls %d (slots refined by teams in %d), mismatches %d, worst relative loss error %.1e, refined pose %.1e, gradient %.1e "
      "(cells a selected hypothesis sampled: %.1e), %.0f s" % (n, teams, bad, worst["loss"], worst["pose"], worst["grad"], worst["grad_sampled"], time.time() - t0))
