import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from esac_amd import api, synthetic as S
eng = api.engine(0)
frames = [S.make_frame(k) for k in range(16)]
for B in (16, 64, 256):
    bc = torch.stack([torch.from_numpy(frames[k % 16]["coords"]) for k in range(B)]).cuda().contiguous()
    ba = torch.stack([torch.from_numpy(S.gating_assignment(frames[k % 16], 256)) for k in range(B)]).cuda().contiguous()
    p = eng.make_params(1, 60, 80, 256)
    eng.set_timing(True)
    ph = np.zeros(6)
    for i in range(12):
        p.call = i * B
        eng.forward_batch(bc, ba, p)
        if i >= 2: ph += eng.phase_ms()
    ph /= 10
    eng.set_timing(False)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(20):
        p.call = i * B
        eng.forward_batch(bc, ba, p)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print("B=%d: wall %.3f ms/batch (%.1f M hyp/s) phases sample %.3f score %.3f select+rescore %.3f refine %.3f total %.3f" % (B, dt * 1e3, B * 256 / dt / 1e6, ph[0], ph[1], ph[2], ph[3], ph[4]))
