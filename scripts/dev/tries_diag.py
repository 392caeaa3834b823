"""A frame of scripts/dev/sweep.py ... multi whose accepted tries differ from the oracle's: which hypotheses, on which expert's map,
what the guaranteed fp64 sampling route says, and the geometry of the minimal set the oracle accepted (the class DESIGN.md's
deviation table calls "ill-conditioned minimal sets").  usage: python scripts/dev/tries_diag.py <base> <key> <k> [<k> ...]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from esac_amd import api, synthetic as S
from oracle import esac_oracle as O

base, key = int(sys.argv[1]), int(sys.argv[2])
eng = api.engine(0)
for k in (int(v) for v in sys.argv[3:]):
    kind = k % 4
    E, N, mode = [(10, 1024, "gating"), (3, 300, "gating"), (12, 2048, "gating"), (5, 700, "dirichlet")][kind]
    f = S.make_frame(base + k, E=E, true_expert=k % E, outlier_frac=0.3 if kind != 1 else 0.55)
    ha = S.gating_assignment(f, N, mode=mode)
    kw = dict(shift_x=f["shift"][0], shift_y=f["shift"][1], focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"], seed=key, call=k)
    sc, hat = torch.from_numpy(f["coords"]).cuda(), torch.from_numpy(ha).cuda()
    ref = O.forward(f["coords"], ha, **kw)
    out = {}
    for name, extra, nospec in (("default (speculative)", {}, False), ("stream order", {}, True), ("exact sampling (fp64, no screen)", {"exact_sampling": True}, True)):
        eng.set_debug(no_speculation=nospec)
        eng.forward_device(sc, hat, eng.make_params(E, 60, 80, N, **kw, **extra))
        out[name] = (eng.read(api.BUF_TRIES).copy(), eng.read(api.BUF_SAMPLE_XY).copy(), eng.read(api.BUF_HYPS).copy())
    eng.set_debug()
    diff = np.nonzero(out["default (speculative)"][0] != ref["tries"])[0]
    print("frame %d (seed %d, E=%d N=%d, true expert %d): winner %d, hypotheses whose accepted try differs: %s" % (k, base + k, E, N, k % E, ref["winner"], diff.tolist()))
    for h in diff:
        xy = ref["sample_xy"][h].reshape(4, 2)
        pts = np.array([[f["coords"][ha[h], c, y, x] for c in range(3)] for x, y in xy], np.float64)
        print("  hypothesis %d on expert %d (%s): oracle try %d" % (h, ha[h], "the true expert" if ha[h] == k % E else "a wrong expert: garbage map", ref["tries"][h]),
              "| " + " | ".join("%s: try %d" % (n, v[0][h]) for n, v in out.items()))
        print("    the oracle's minimal set: cells %s, pixel extent %d x %d px, scene extent %.2f m, depth of the oracle's pose %.1f m; is the winner: %s"
              % (xy.tolist(), 8 * (xy[:, 0].max() - xy[:, 0].min()), 8 * (xy[:, 1].max() - xy[:, 1].min()), np.linalg.norm(pts.max(0) - pts.min(0)),
                 ref["hyps"][h][5] if "hyps" in ref else float("nan"), h == ref["winner"]))
