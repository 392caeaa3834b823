import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from esac_amd import api, synthetic as S
for cfg in ((10, 1024), (12, 4096)):
    E, N = cfg
    eng = api.Engine(0)
    frames = [S.make_frame(k, E=E) for k in range(8)]
    sc = [torch.from_numpy(f["coords"]).cuda() for f in frames]
    ha = [torch.from_numpy(S.gating_assignment(f, N, mode="gating")).cuda() for f in frames]
    p = eng.make_params(E, 60, 80, N, seed=1320, call=0)
    for i in range(200):
        p.call = i
        eng.forward_device(sc[i % 8], ha[i % 8], p)
    c = eng.read(api.BUF_CYCLES)
    n = max(1, c[30])
    print(cfg, "join sections (us): wait %.2f | loads+max %.2f | classify %.2f | stats %.2f | pick+deliver %.2f  (n=%d)" % tuple(list(c[24:29] / n * 0.01) + [n]))
