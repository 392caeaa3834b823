import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from esac_amd import api, synthetic as S
dev = torch.device("cuda", 0)
frames = [S.make_frame(k, E=1, H=60, W=80, sub=8) for k in range(16)]
assigns = [S.gating_assignment(f, 256, mode="single") for f in frames]
eng = api.engine(0)
kw = dict(focal=frames[0]["focal"], ppx=frames[0]["ppx"], ppy=frames[0]["ppy"], sub_sampling=8)
d_assign = [torch.from_numpy(a).to(dev) for a in assigns]
d_coords = [torch.from_numpy(f["coords"]).to(dev) for f in frames]
scores = torch.empty(256, dtype=torch.float64, device=dev)
params = eng.make_params(1, 60, 80, 256, seed=1305, call=0, **kw)
ts = []
torch.cuda.synchronize()
for i in range(120):
    params.call = i
    t0 = time.perf_counter()
    r = eng.forward_device(d_coords[i % 16], d_assign[i % 16], params, scores_out=scores)
    ts.append((time.perf_counter() - t0) * 1e3)
print("per-call ms:", " ".join("%.3f" % t for t in ts[:40]))
print("mean 5..25: %.4f  mean 40..120: %.4f" % (np.mean(ts[5:25]), np.mean(ts[40:])))
time.sleep(2.0)
ts2 = []
for i in range(40):
    params.call = 200 + i
    t0 = time.perf_counter()
    r = eng.forward_device(d_coords[i % 16], d_assign[i % 16], params, scores_out=scores)
    ts2.append((time.perf_counter() - t0) * 1e3)
print("after 2 s idle:", " ".join("%.3f" % t for t in ts2[:24]))
