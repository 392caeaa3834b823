import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from esac_amd import api, synthetic as S
eng = api.engine(0)
mrs = int(sys.argv[1])
frames = [S.make_frame(k) for k in range(8)]
sc = [torch.from_numpy(f["coords"]).cuda() for f in frames]
ha = [torch.from_numpy(S.gating_assignment(f, 256)).cuda() for f in frames]
p = eng.make_params(1, 60, 80, 256, max_ref_steps=mrs)
steps = lm = 0
for i in range(120):
    p.call = i
    r = eng.forward_device(sc[i % 8], ha[i % 8], p)
    steps += r[api.RES_REF_STEPS]; lm += r[api.RES_LM_ITERS]
print("max_ref_steps", mrs, "accepted steps/frame", steps / 120, "lm iters/frame", lm / 120)
