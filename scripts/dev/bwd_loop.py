"""Backward calls of the default workload (cfg2: 1 expert, 256 hyps, 60x80) for rocprofv3 --kernel-trace --stats."""
import sys
import numpy as np, torch
import os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from esac_amd import api, synthetic as S
alpha = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
eng = api.engine(0)
frames = [S.make_frame(k) for k in range(8)]
sc = [torch.from_numpy(f["coords"]).cuda() for f in frames]
ha = [torch.from_numpy(S.gating_assignment(f, 256)).cuda() for f in frames]
g = torch.zeros_like(sc[0])
slots = 0
for i in range(60):
    k = i % 8
    g.zero_()
    o = eng.backward_device(sc[k], g, ha[k], frames[k]["gt_pose"].astype(np.float32), 1.0, 100.0, 100.0,
                            eng.make_params(1, 60, 80, 256, seed=1305, call=i, inlier_alpha=alpha))
    slots += o[1]
print("alpha", alpha, "mean slots", slots / 60, "loss", o[0])
