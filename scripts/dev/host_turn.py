"""round 5: where the host's share of a headline step goes (the "host turn": ms_per_step - sum of the kernels' durations).
Per step of the bench's own loop (cfg2, device-resident inputs): Python before the C call, inside esac_hip_forward
(esac_hip_host_turn: argument block, the three launch calls, record landed, return), Python after it.  Medians over the steps."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from esac_amd import api, synthetic as S
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
eng = api.engine(0)
frames = [S.make_frame(k) for k in range(16)]
assigns = [S.gating_assignment(f, 256, mode="single") for f in frames]
d_coords = [torch.from_numpy(f["coords"]).cuda() for f in frames]
d_assign = [torch.from_numpy(a).cuda() for a in assigns]
scores = torch.empty(256, dtype=torch.float64, device="cuda")
params = eng.make_params(1, 60, 80, 256, seed=1320, call=0, focal=frames[0]["focal"], ppx=frames[0]["ppx"], ppy=frames[0]["ppy"], sub_sampling=8)
rows = []
for i in range(40 + steps):
    k = i % 16
    params.call = i
    t0 = time.perf_counter_ns()
    eng.forward_device(d_coords[k], d_assign[k], params, scores_out=scores)
    t1 = time.perf_counter_ns()
    h = eng.host_turn()
    if i >= 40:
        rows.append((t0, t1, h))
    # (the host_turn() read itself is outside [t0, t1] but inside the loop: the between-calls figure below contains it)
t_all = (rows[-1][1] - rows[0][0]) / len(rows) * 1e-3
med = lambda v: float(np.median(v))
pre = med([(h["entry_ns"] - t0) * 1e-3 for t0, t1, h in rows])
args = med([h["args_ready"] for _, _, h in rows])
l1 = med([h["sample_launched"] - h["args_ready"] for _, _, h in rows])
l2 = med([h["score_launched"] - h["sample_launched"] for _, _, h in rows])
l3 = med([h["refine_launched"] - h["score_launched"] for _, _, h in rows])
wait = med([h["record_landed"] - h["refine_launched"] for _, _, h in rows])
tail = med([h["returned"] - h["record_landed"] for _, _, h in rows])
post = med([(t1 - h["entry_ns"]) * 1e-3 - h["returned"] for t0, t1, h in rows])
between = med([(rows[i + 1][0] - rows[i][1]) * 1e-3 for i in range(len(rows) - 1)])
call = med([(t1 - t0) * 1e-3 for t0, t1, h in rows])
print("steps %d: %.2f us per step (this loop, incl. the host_turn read)" % (len(rows), t_all))
print("  Python: forward_device entry -> C entry           %6.2f us" % pre)
print("  C: validation + argument block (make_args)         %6.2f us" % args)
print("  C: launch k_sample                                 %6.2f us" % l1)
print("  C: launch k_score_fast                             %6.2f us" % l2)
print("  C: launch k_refine_team                            %6.2f us" % l3)
print("  C: polling until the record has landed             %6.2f us   (the GPU works: ~ sum of the kernels - what the launches above overlapped)" % wait)
print("  C: record landed -> return                         %6.2f us" % tail)
print("  Python: C return -> forward_device returned        %6.2f us" % post)
print("  Python: between two forward_device calls           %6.2f us   (this loop; the bench's own loop is measured by bench.py)" % between)
print("  one forward_device call                            %6.2f us" % call)
