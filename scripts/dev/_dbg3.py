import os, sys, time, socket
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch.distributed as dist
from esac_amd import api, distributed as D, synthetic as S
eng = api.engine(0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
f = S.make_frame(0); ha = torch.from_numpy(S.gating_assignment(f, 256)).cuda(); sc = torch.from_numpy(f["coords"]).cuda()
kw = dict(seed=1320, exact_scores="auto", focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=8)
for i in range(20):
    D.forward_sharded(eng, sc, ha, dict(kw, call=i), policy="range")
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for i in range(200):
    D.forward_sharded(eng, sc, ha, dict(kw, call=20 + i), policy="range")
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 200
pr.disable()
print("per call %.1f us" % (dt * 1e6))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
dist.destroy_process_group()
