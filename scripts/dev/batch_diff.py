import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from esac_amd import api, synthetic as S
eng = api.engine(0)
N = 96
for B in (12,):
    frames = [S.make_frame(70 + b, E=2, true_expert=b % 2) for b in range(B)]
    assigns = np.stack([S.gating_assignment(f, N, mode="gating") for f in frames])
    coords = torch.from_numpy(np.stack([f["coords"] for f in frames])).cuda()
    ha = torch.from_numpy(assigns).cuda()
    p = eng.make_params(2, 60, 80, N, call=40)
    res_b = eng.forward_batch(coords, ha, p)
    res_b2 = eng.forward_batch(coords, ha, p)
    print("batch repeat equal:", np.array_equal(res_b, res_b2))
    for b in range(B):
        q = eng.make_params(2, 60, 80, N, call=40 + b)
        r1 = eng.forward_device(coords[b], ha[b], q)
        r2 = eng.forward_device(coords[b], ha[b], q)
        d = np.nonzero(res_b[b][:31] != r1[:31])[0]
        print(b, "single repeat equal", np.array_equal(r1, r2), "diff idx", d, [(res_b[b][i], r1[i]) for i in d][:4], "steps", r1[25], res_b[b][25], "lm", r1[30], res_b[b][30])
