"""Which gradient path of which slot deviates from the oracle on a frame of bwd_sweep.py: python scripts/dev/bwd_diag.py <k> ..."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from esac_amd import api, synthetic as S  # noqa: E402
from oracle import esac_oracle as O  # noqa: E402

eng = api.Engine(0)
for k in [int(v) for v in sys.argv[1:]]:
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
    g_ref = np.zeros_like(f["coords"])
    ref = O.backward(f["coords"], g_ref, ha, gt, w_rot=1.0, w_trans=100.0, loss_cut=100.0, focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"],
                     sub_sampling=f["sub"], inlier_alpha=alpha, seed=55, call=k, want_paths=True)
    sel = np.nonzero(ref["probs"] >= 1e-3)[0]
    n = int(out[1])
    p1 = eng.read_slabs(api.BUF_BWD_PATH1, n).reshape(n, 3, -1)   # [slot,3,P]
    p2 = eng.read_slabs(api.BUF_BWD_PATH2, n).reshape(n, 3, -1)
    slots = eng.read(api.BUF_BWD_SLOTS)[:n]
    info = eng.read(api.BUF_BWD_SLOT_INFO)[:n]
    scale = np.abs(g_ref).max()
    print("frame %d: E %d N %d alpha %g slots %d, |grad|max %.3g, total grad err %.2e" % (k, E, N, alpha, n, scale, np.abs(g.cpu().numpy() - g_ref).max() / scale))
    for s_, h in enumerate(slots):
        r1 = ref["grad_path1"][h].T  # [3,P]
        r2 = ref["grad_path2"][h].T
        # both sides index cells as y * W + x (oracle/esac_oracle_bwd.inc, the assemble loop)
        pr = ref["probs"][h]
        d1 = np.abs(p1[s_] - r1).max() * pr / scale
        d2 = np.abs(p2[s_] - r2).max() / scale
        if d1 > 2e-7 or d2 > 2e-7:
            print("   slot %2d hyp %3d prob %.3g  path I err %.2e (|pI| %.3g, oracle %.3g)  path II err %.2e (|pII| %.3g)  steps %d inliers %d" % (
                s_, h, pr, d1, np.abs(p1[s_]).max(), np.abs(r1).max(), d2, np.abs(p2[s_]).max(), info[s_][2], info[s_][1]))
