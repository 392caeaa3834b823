import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from esac_amd import api, synthetic as S
eng = api.engine(0)
f = S.make_frame(0, H=12, W=16, sub=40)
ha = torch.from_numpy(S.gating_assignment(f, 8)).cuda()
sc = torch.from_numpy(f["coords"]).cuda()
p = eng.make_params(1, 12, 16, 8, sub_sampling=40, max_ref_steps=0)
for i in range(200): eng.forward_device(sc, ha, p)
torch.cuda.synchronize(); t=time.perf_counter()
for i in range(3000):
    p.call = i; eng.forward_device(sc, ha, p)
torch.cuda.synchronize(); dt = time.perf_counter()-t
print("tiny problem, blocking call: %.1f us per call" % (dt/3000*1e6))
t=time.perf_counter()
for i in range(3000):
    p.call = i; eng.forward_device(sc, ha, p, want_host=False)
torch.cuda.synchronize(); dt = time.perf_counter()-t
print("tiny problem, asynchronous calls back to back: %.1f us per call" % (dt/3000*1e6))
import ctypes as C
t=time.perf_counter()
for i in range(3000):
    p.call = i
print("python loop + attribute set only: %.2f us" % ((time.perf_counter()-t)/3000*1e6))
