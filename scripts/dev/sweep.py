import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from esac_amd import api, synthetic as S
from oracle import esac_oracle as O
eng = api.engine(0)
n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
base = int(sys.argv[2]) if len(sys.argv) > 2 else 1000   # first frame seed
key = int(sys.argv[3]) if len(sys.argv) > 3 else 77       # RNG seed of the calls
multi = len(sys.argv) > 4 and sys.argv[4] == "multi"       # several experts, 300..2048 hypotheses: the speculative route (round 6)
spec_calls0 = eng.spec_info()["calls"]
bad = 0; worst_r = worst_t = 0.0; flips = 0; lm_diff = 0
t0 = time.time()
for k in range(n_frames):
    kind = k % 4
    if multi:
        E, N, mode = [(10, 1024, "gating"), (3, 300, "gating"), (12, 2048, "gating"), (5, 700, "dirichlet")][kind]
        f = S.make_frame(base + k, E=E, true_expert=k % E, outlier_frac=0.3 if kind != 1 else 0.55)
    elif kind == 0: f = S.make_frame(base + k); N = 256; mode = "single"
    elif kind == 1: f = S.make_frame(base + k, E=3, true_expert=k % 3); N = 192; mode = "gating"
    elif kind == 2: f = S.make_frame(base + k, noise=0.05, outlier_frac=0.5); N = 128; mode = "single"
    else: f = S.make_frame(base + k, H=45, W=61, sub=10, shift=(k % 7 - 3, 2)); N = 96; mode = "single"
    ha = S.gating_assignment(f, N, mode=mode)
    E, _, H, W = f["coords"].shape
    p = eng.make_params(E, H, W, N, shift_x=f["shift"][0], shift_y=f["shift"][1], focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"],
                        sub_sampling=f["sub"], seed=key, call=k)
    res = eng.forward_device(torch.from_numpy(f["coords"]).cuda(), torch.from_numpy(ha).cuda(), p)
    ref = O.forward(f["coords"], ha, shift_x=f["shift"][0], shift_y=f["shift"][1], focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"],
                    sub_sampling=f["sub"], seed=key, call=k)
    ok = int(res[api.RES_HYP]) == ref["winner"] and int(res[api.RES_REF_STEPS]) == ref["ref_steps"]
    if multi:  # the straggler chain ran beside the refinement: its accepted tries and cells as well
        ok = ok and np.array_equal(eng.read(api.BUF_TRIES), ref["tries"]) and np.array_equal(eng.read(api.BUF_SAMPLE_XY), ref["sample_xy"])
    cnt_equal = np.array_equal(eng.read(api.BUF_INLIER_COUNTS), ref["inlier_counts"])
    map_equal = np.array_equal(eng.read(api.BUF_INLIER_MAP), ref["inlier_map"])
    r, t = S.pose_errors(res[api.RES_POSE:api.RES_POSE + 16].reshape(4, 4), ref["pose"])
    worst_r, worst_t = max(worst_r, r), max(worst_t, t)
    lm_diff += int(res[api.RES_LM_ITERS]) != ref["lm_iters"]
    if not (ok and cnt_equal and map_equal and r <= 1e-4 and t <= 1e-3):
        bad += 1
        print("MISMATCH frame", k, "kind", kind, "winner", int(res[api.RES_HYP]), ref["winner"], "steps", int(res[api.RES_REF_STEPS]), ref["ref_steps"],
              "counts", cnt_equal, "map", map_equal, "r", r, "t", t)
print("frames %d (seeds %d.., key %d%s) mismatches %d worst rot %.2e rad worst trans %.2e m, LM-iteration count differs on %d frames, %.1f s" % (n_frames, base, key, ", several experts" if multi else "", bad, worst_r, worst_t, lm_diff, time.time() - t0))
if multi:
    si = eng.spec_info()
    print("speculative calls %d of %d, speculations that failed (a straggler won: refined again) %d" % (si["calls"] - spec_calls0, n_frames, si["failures"]))
