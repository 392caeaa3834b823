"""round 5: is the 66 / 67 / 71 us process-to-process state of the team refinement the PLACEMENT of its exchange granules
(which memory channel / stack the 2 x 4 KB of an exchange live on, near or far from the XCD the team runs on)?  A probe build
of the library (scratch/lib_granshift.so: esac_probe_gran_shift moves the granule base in units of 128 bytes inside an
over-allocated buffer) -- refine stage time against the shift, inside ONE process."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from esac_amd import api, synthetic as S
eng = api.engine(0)
shift = C.c_int.in_dll(eng.lib, "esac_probe_gran_shift")
f = S.make_frame(3)
ha = S.gating_assignment(f, 256, mode="single")
sc, hat = torch.from_numpy(f["coords"]).cuda(), torch.from_numpy(ha).cuda()
p = eng.make_params(1, 60, 80, 256, seed=1320, call=5, exact_scores="auto")
for k in range(5):
    eng.forward_device(sc, hat, p)

def refine_us(units, reps=20):
    shift.value = units
    return eng.time_stages(sc, hat, p, reps=reps)["refine"] * 1e3

print("census", eng.refine_info()["xcd_census"])
base = [refine_us(0) for _ in range(5)]
print("shift 0, five times: " + " ".join("%.2f" % v for v in base))
print("-- pages (4 KB steps), 0..255: refine us")
pages = [refine_us(pg * 32) for pg in range(256)]
for r in range(0, 256, 16):
    print("%3d: " % r + " ".join("%5.1f" % v for v in pages[r:r + 16]))
print("-- 128-byte steps inside the first 8 KB")
fine = [refine_us(u) for u in range(64)]
for r in range(0, 64, 16):
    print("%3d: " % r + " ".join("%5.1f" % v for v in fine[r:r + 16]))
print("-- 64 KB steps, 0..15")
big = [refine_us(k * 512) for k in range(16)]
print(" ".join("%5.1f" % v for v in big))
best, worst = int(np.argmin(pages)), int(np.argmax(pages))
print("best page %d (%.2f us), worst page %d (%.2f us); again: %.2f / %.2f" % (best, pages[best], worst, pages[worst], refine_us(best * 32), refine_us(worst * 32)))
for name, pg in (("best", best), ("worst", worst), ("zero", 0)):
    shift.value = pg * 32
    for i in range(40):
        p.call = i
        eng.forward_device(sc, hat, p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(400):
        p.call = 40 + i
        eng.forward_device(sc, hat, p)
    torch.cuda.synchronize()
    print("%s page %d: %.2f us per blocking call, census %s" % (name, pg, (time.perf_counter() - t0) / 400 * 1e6, eng.refine_info()["xcd_census"]))
