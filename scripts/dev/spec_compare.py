"""Speculative forward (straggler chain beside the refinement) against the serial route on the same inputs: every stage buffer
and the record must be identical.  All serial calls first, then the speculative ones in ANOTHER order (what a call finds in the
workspace is then another frame's state, not its own serial twin's).  usage: python scripts/dev/spec_compare.py [frames]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from esac_amd import api, synthetic as S

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng = api.engine(0)
KEYS = ("tries", "xy", "hyps", "flags", "scores", "user", "counts", "imap")


def run(E, N, mode, k, nospec):
    f = S.make_frame(100 + k, E=E)
    ha = S.gating_assignment(f, N, mode=mode)
    sc, hat = torch.from_numpy(f["coords"]).cuda(), torch.from_numpy(ha).cuda()
    eng.set_debug(no_speculation=nospec)
    p = eng.make_params(E, 60, 80, N, seed=1305, call=k)
    scores = torch.zeros(N, dtype=torch.float64, device="cuda")
    rec = eng.forward_device(sc, hat, p, scores_out=scores)
    return dict(rec=rec.copy(), hyps=eng.read(api.BUF_HYPS), tries=eng.read(api.BUF_TRIES), xy=eng.read(api.BUF_SAMPLE_XY),
                scores=eng.read(api.BUF_SCORES), flags=eng.read(api.BUF_EXACT_FLAGS), user=scores.cpu().numpy(),
                counts=eng.read(api.BUF_INLIER_COUNTS), imap=eng.read(api.BUF_INLIER_MAP), info=eng.spec_info(),
                stragglers=int(eng.read(api.BUF_SPEC_FLAGS).sum()) if not nospec else 0)


bad = 0
for (E, N, mode) in ((3, 300, "gating"), (2, 500, "gating"), (10, 1024, "gating"), (12, 4096, "gating"), (6, 1500, "dirichlet"), (4, 8192, "gating")):
    serial = {k: run(E, N, mode, k, True) for k in range(n_frames)}
    fails = strag = 0
    for k in list(range(n_frames))[::-1]:
        a, b = serial[k], run(E, N, mode, k, False)
        assert not a["info"]["last_speculative"] and b["info"]["last_speculative"], (a["info"], b["info"])
        fails += b["info"]["last_failed"]
        strag += b["stragglers"]
        for key in KEYS:
            if not np.array_equal(a[key], b[key], equal_nan=True):
                d = np.nonzero(np.asarray(a[key] != b[key]).reshape(len(a[key]), -1).any(axis=1))[0]
                print("MISMATCH E=%d N=%d frame %d: %s differs at %d entries, first %s: serial %s spec %s" % (E, N, k, key, len(d), d[:5], a[key][d[0]], b[key][d[0]]))
                bad += 1
        ra, rb = a["rec"], b["rec"]
        if not np.array_equal(ra[:31], rb[:31]):
            d = np.nonzero(ra[:31] != rb[:31])[0]
            print("MISMATCH E=%d N=%d frame %d: record fields %s: serial %s spec %s" % (E, N, k, d, ra[d], rb[d]))
            bad += 1
    print("E=%d N=%d %s: %d frames compared, %.1f stragglers per frame, speculation failed on %d" % (E, N, mode, n_frames, strag / n_frames, fails))
eng.set_debug()
print("mismatches:", bad)
