"""round 5: what a forward call does while another kernel holds the CUs of XCD 0 (tests/native/filler.hip), for several
leave intervals of the filler's workgroups: call duration, how the refinement ran, team fall-backs."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from esac_amd import api, synthetic as S
from tests.native import build as nb
lib = C.CDLL(nb.build_filler())
lib.filler_launch.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
f = S.make_frame(433)
ha = S.gating_assignment(f, 128)
sc, hat = torch.from_numpy(f["coords"]).cuda(), torch.from_numpy(ha).cuda()
side = torch.cuda.Stream()
counter = torch.zeros(1, dtype=torch.int32, device="cuda")
for step in (0.1, 0.25, 0.5, 1.0, 2.0):
    for blocks in (256, 512):
        eng = api.Engine(0)
        p = eng.make_params(1, 60, 80, 128, seed=29, call=2)
        eng.forward_device(sc, hat, p)
        torch.cuda.synchronize()
        lib.filler_launch(C.c_void_p(side.cuda_stream), 0, step, blocks, C.c_void_p(counter.data_ptr()))
        time.sleep(0.001)
        t0 = time.perf_counter()
        eng.forward_device(sc, hat, p)
        dt = time.perf_counter() - t0
        info = eng.refine_info()
        torch.cuda.synchronize()
        tf = time.perf_counter() - t0
        print("step %.2f ms blocks %d: call %.3f ms (filler done after %.3f ms, %d on XCD 0) mode %s fallbacks %d timed_out %s exchanges %d census %s" %
              (step, blocks, dt * 1e3, tf * 1e3, int(counter.item()), info["mode"], info["team_fallbacks"], info["timed_out"], info["exchanges"], info["xcd_census"]))
