import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from esac_amd import api, distributed as D, synthetic as S
engine = api.engine(0)
world = 2
f = S.make_frame(61, E=2, true_expert=1)
ha = S.gating_assignment(f, 250, mode="gating")
sc = torch.from_numpy(f["coords"]).cuda(); hat = torch.from_numpy(ha).cuda()
kw = dict(seed=1305, call=9)
n_total = 250
def run(nospec):
    engine.set_debug(no_speculation=nospec)
    out = []
    for rank in range(world):
        buf = torch.zeros(n_total + world * D.RES_DOUBLES, dtype=torch.float64, device="cuda")
        D.contribute_range(engine, sc, hat, kw, rank, world, buf)
        torch.cuda.synchronize()
        out.append(dict(tries=engine.read(api.BUF_TRIES), hyps=engine.read(api.BUF_HYPS), scores=engine.read(api.BUF_SCORES), xy=engine.read(api.BUF_SAMPLE_XY),
                        user=buf.cpu().numpy()))
    return out
ref = run(True)
for rep in range(3):
    got = run(False)
    for rank in range(world):
        for key in ("tries", "xy", "hyps", "scores"):
            a, b = ref[rank][key], got[rank][key]
            d = np.nonzero((a != b).reshape(len(a), -1).any(axis=1))[0]
            if len(d):
                print("rep", rep, "rank", rank, key, "differs at", d[:8], "serial", a[d[0]], "spec", b[d[0]])
print("done")
print("serial rank1 local 76 score", ref[1]["scores"][76], "serial rank0 76", ref[0]["scores"][76], "spec rank0 76", got[0]["scores"][76], "spec rank1 76", got[1]["scores"][76])
print("tries rank0 76", ref[0]["tries"][76], "rank1 76", ref[1]["tries"][76])
# long stragglers of each shard and whether their scores match
for rank in range(world):
    t = ref[rank]["tries"]; idx = np.nonzero(t >= 32)[0]
    print("rank", rank, "stragglers", [(int(i), int(t[i]), float(ref[rank]["scores"][i]), float(got[rank]["scores"][i])) for i in idx])
# recompute the fp32 scores of rank 0's shard from the workspace as the speculative call left it
engine.set_debug(no_speculation=False)
buf = torch.zeros(n_total + world * D.RES_DOUBLES, dtype=torch.float64, device="cuda")
D.contribute_range(engine, sc, hat, kw, 0, world, buf)
torch.cuda.synchronize()
s_spec = engine.read(api.BUF_SCORES)[76]
p = engine.make_params(2, 60, 80, 125, hyp_offset=0, **kw)
engine.score(sc, hat[:125], p); engine.select(sc, hat[:125], p)
torch.cuda.synchronize()
print("spec call's score of 76:", s_spec, "recomputed from its workspace:", engine.read(api.BUF_SCORES)[76])
buf = torch.zeros(n_total + world * D.RES_DOUBLES, dtype=torch.float64, device="cuda")
D.contribute_range(engine, sc, hat, kw, 0, world, buf)
torch.cuda.synchronize()
fl = engine.read(api.BUF_SPEC_FLAGS)
print("spec flags set:", np.nonzero(fl)[0], "tries there", engine.read(api.BUF_TRIES)[np.nonzero(fl)[0]])
