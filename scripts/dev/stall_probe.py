import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from esac_amd import api, synthetic as S
f = S.make_frame(433)
ha = S.gating_assignment(f, 128)
sc, hat = torch.from_numpy(f["coords"]).cuda(), torch.from_numpy(ha).cuda()
eng = api.Engine(0)
p = eng.make_params(1, 60, 80, 128, seed=29, call=2)
for i in range(3):
    eng.forward_device(sc, hat, p)
eng.set_debug(coop_stall=True)
for i in range(4):
    t0 = time.perf_counter()
    eng.forward_device(sc, hat, p)
    dt = time.perf_counter() - t0
    print("stalled call %d: %.3f ms %s" % (i, dt * 1e3, eng.refine_info()))
