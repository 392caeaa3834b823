#!/usr/bin/env python
"""bench.py -- pose hypotheses/sec of the ESAC hot path (`esac.forward`) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched by
torch.distributed.run, one rank per GPU, RCCL.  W untimed warm-up steps, EXACTLY K timed
steps bracketed by barrier + torch.cuda.synchronize(), MAX over ranks, rank 0 prints ONE
JSON line.

A "step" = one complete pass of the hot path over one frame: sample+P3P -> soft-inlier
score of every hypothesis over the whole coordinate grid -> select -> refine the winner
-> 4x4 pose on the host.  Inputs (scene-coordinate maps, assignment vector) are already
resident in HBM when the timed region starts.  Workload at N=1: BASELINE.json configs[1]
("7-Scenes chess", 1 expert, 256 hypotheses, 640x480 frame -> 60x80 grid), synthetic data.
For N>1 every rank scores its own 256 hypotheses of the SAME frame (weak scaling: the
global hypothesis count grows with N) and one all-reduce on the score vector + candidate
records picks the global winner.

`roofline`: the dominant streaming kernel (k_score_fast).  achieved = algorithmic bytes per
launch (N * 12*H*W, SURVEY.md 8d: every hypothesis reads x,y,z of its expert's map once) /
mean launch duration, measured with hipEvents recorded on the launch stream around that
kernel inside the timed region.  `cpu_baseline`: the CPU oracle (a port: the reference
needs OpenCV and cannot be built here) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from esac_amd import api, synthetic as S  # noqa: E402
from esac_amd import distributed as D  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(frames, assigns, n_hyp):
    """Oracle timed on the host cores, bounded sample (~10-30 s). Only the checker's timing leg uses oracle/."""
    from oracle import esac_oracle as O
    best = None
    for threads in sorted({1, O.max_threads()}):
        t_budget = time.time()
        times = []
        for i in range(2 + 12):
            f, ha = frames[i % len(frames)], assigns[i % len(frames)]
            t0 = time.time()
            O.forward(f["coords"], ha, focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"],
                      seed=1305, call=i, num_threads=threads)
            dt = time.time() - t0
            if i >= 2:
                times.append(dt)
            if time.time() - t_budget > 15.0 and len(times) >= 3:
                break
        med = float(np.median(times))
        if best is None or med < best[0]:
            best = (med, threads, len(times))
    med, threads, n = best
    out = {"value": n_hyp / med, "unit": "hypotheses/s", "cores": threads, "kind": "port",
           "sample": "median of %d oracle esac_forward calls on the same workload (%d hyp, 60x80 grid), %d thread(s); "
                     "host has %d hardware threads" % (n, n_hyp, threads, os.cpu_count())}
    # oracle/_ref = the reference's own esac_util.h code (OpenCV stand-in shim), its OpenMP pragmas on all threads
    try:
        from oracle import ref_binding
        if os.path.exists(ref_binding.LIB_PATH):
            import ctypes as C
            L = ref_binding.lib()
            times = []
            f, ha = frames[0], np.ascontiguousarray(assigns[0], np.int64)
            sc = np.ascontiguousarray(f["coords"], np.float32)
            E, _, H, W = sc.shape
            bufs = [np.zeros((4, 4), np.float32), np.zeros((n_hyp, 8), np.int32), np.zeros((n_hyp, 6)), np.zeros(n_hyp),
                    np.zeros(1, np.int32), np.zeros(6), np.zeros((H, W), np.uint8), np.zeros(1)]
            p = lambda a: a.ctypes.data_as(C.c_void_p)
            for i in range(2 + 10):
                t0 = time.time()
                L.ref_forward(p(sc), E, H, W, p(ha), n_hyp, p(bufs[0]), 0, 0, f["focal"], f["ppx"], f["ppy"], 10.0, 100.0,
                              0.5, 100.0, f["sub"], 1000000, 100, *[p(b) for b in bufs[1:]])
                if i >= 2:
                    times.append(time.time() - t0)
            ref_rate = n_hyp / float(np.median(times))
            out["reference_sources_value"] = ref_rate
            out["sample"] += "; oracle/_ref (reference esac_util.h + OpenCV stand-in, all OpenMP threads): %.0f hypotheses/s" % ref_rate
            if ref_rate > out["value"]:
                out.update(value=ref_rate, kind="reference", cores=O.max_threads())
    except Exception as exc:  # the baseline leg must never break the bench line
        out["sample"] += "; oracle/_ref not timed (%s)" % type(exc).__name__
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--hyps", type=int, default=256, help="hypotheses per GPU (configs[1]: 256)")
    ap.add_argument("--experts", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--grid", type=str, default="60x80")
    ap.add_argument("--batch", type=int, default=64, help="frames per launch set for the extra `batched` figure (0 = skip)")
    ap.add_argument("--no-training", action="store_true", help="skip the extra `training` (esac.backward) figure")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    # test hook (tests/test_gpu_distributed.py): several ranks on ONE device with gloo, to exercise this file's
    # multi-rank path on a single-GPU box; RCCL itself refuses two ranks on one device
    one_device = os.environ.get("ESAC_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
    assert args.gpus == world or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"

    H, W = (int(v) for v in args.grid.split("x"))
    sub = 8 if (H, W) == (60, 80) else max(1, 480 // H)
    n_frames = 16
    n_local = args.hyps
    n_total = n_local * world
    frames = [S.make_frame(k, E=args.experts, H=H, W=W, sub=sub) for k in range(n_frames)]
    assigns = [S.gating_assignment(f, n_total, mode="single" if args.experts == 1 else "gating") for f in frames]
    eng = api.engine(local_rank)
    d_coords = [torch.from_numpy(f["coords"]).to(dev) for f in frames]
    d_assign = [torch.from_numpy(a).to(dev) for a in assigns]
    kw = dict(focal=frames[0]["focal"], ppx=frames[0]["ppx"], ppy=frames[0]["ppy"], sub_sampling=sub)
    scores = torch.empty(n_local, dtype=torch.float64, device=dev)
    PHASE_EVERY = 16  # the phase events and device-side stamps themselves cost GPU time: sample every 16th step

    params = eng.make_params(args.experts, H, W, n_local, seed=1305, call=0, **kw)  # per step only the call counter moves

    def step(i):
        k = i % n_frames
        if world == 1:
            params.call = i
            res = eng.forward_device(d_coords[k], d_assign[k], params, scores_out=scores)
            return res
        _, rec = D.forward_sharded(eng, d_coords[k], d_assign[k], dict(seed=1305, call=i, **kw), policy="range")
        return rec

    def sync():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    phase = np.zeros(6, np.float64)
    n_phase = 0
    lm_iters = ref_steps = 0.0
    eng.set_timing(True, period=PHASE_EVERY)  # the first timed step is a sampled one
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        r = step(args.warmup + i)
        if i % PHASE_EVERY == 0:  # this step recorded its phase events
            phase += eng.phase_ms()
            n_phase += 1
        lm_iters += r[api.RES_LM_ITERS]
        ref_steps += r[api.RES_REF_STEPS]
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_device else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    phase /= max(n_phase, 1)
    span_ms, span_n = eng.score_span_ms()

    default_workload = (args.experts, n_local, H, W) == (1, 256, 60, 80)
    if rank == 0:
        # Duration of the score kernel: device-side span (max end - min start over its workgroups, 100 MHz
        # wall clock), averaged over every launch since timing was enabled (warm-up + timed steps).  The
        # hipEvent bracket around the same launch (phase_ms.score_bracketed) also contains the launch gap.
        score_ms = max(span_ms, 1e-6)
        alg_bytes = n_local * 12.0 * H * W
        achieved = alg_bytes / (score_ms * 1e-3) / 1e9 if score_ms > 0 else 0.0
        out = {
            "metric": "pose hypotheses/sec @640x480, 256 hyp",
            "value": n_total * args.steps / elapsed,
            "unit": "hypotheses/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64 (P3P, LM refinement, exact decisions) + f32 (streaming soft-inlier score, ranking only)",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 1 expert, %d hypotheses/GPU, %dx%d grid (640x480 frame, "
                                   "sub-sampling %d), tau=10 alpha=100 beta=0.5 maxReproj=100; box-room frames, "
                                   "2 cm noise, 30%% outliers" % (n_local, H, W, sub),
                       "experts": args.experts, "hypotheses_per_gpu": n_local, "hypotheses_total": n_total,
                       "grid": [H, W], "frames_cycled": n_frames,
                       "parallelism": "hypotheses sharded over %d GPU(s); 1 all-reduce(SUM) of N+32*world doubles" % world},
            "phase_ms": {"sample_p3p": float(phase[0]), "score": score_ms, "select_rescore": float(phase[2]),
                         "refine": float(phase[3]), "gpu_total": float(phase[4]), "event_bracket_overhead": float(phase[5]),
                         "score_bracketed": float(phase[1]),
                         "refine_steps_per_frame": ref_steps / args.steps, "lm_iters_per_frame": lm_iters / args.steps},
            "roofline": {"kernel": "k_score_fast", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE 272.0 KB
                         # doubled per the gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE 8.6 KB); only valid
                         # for the default workload the profile was taken on, null otherwise
                         "traffic": (2 * 272.02 + 8.58) * 1024 if default_workload else None,
                         "traffic_source": "profiles/r01_bench_cfg2_rocprofv3_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)",
                         # the same launch as rocprofv3's kernel trace times it (dispatch to completion signal, which for
                         # a ~3 us kernel adds the command processor's launch and end-of-kernel cache work): 4.51 us in
                         # the committed summary.  `achieved` above uses the device-side span; both are given.
                         "rocprofv3_kernel_ms": 0.00451 if default_workload else None,
                         "achieved_at_rocprofv3_duration": alg_bytes / 0.00451e-3 / 1e9 if default_workload else None,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel_ms": score_ms,
                         "note": "60x80 grid: 14.7 MB algorithmic per launch, map re-read from L2 by every "
                                 "hypothesis -> latency-bound, see DESIGN.md"},
        }
        if args.batch > 0 and world == 1:
            # extra figure (not `value`): B independent frames per launch set through esac_hip_forward_batch --
            # the single call's tail is one CU of fp64 work, so frames in flight are what fills the chip
            Bf = args.batch
            eng.set_timing(False)
            bc = torch.stack([d_coords[k % n_frames] for k in range(Bf)]).contiguous()
            ba = torch.stack([d_assign[k % n_frames][:n_local] for k in range(Bf)]).contiguous()
            bscores = torch.empty(Bf, n_local, dtype=torch.float64, device=dev)
            nb = max(4, min(40, args.steps // 8))
            for i in range(3):
                eng.forward_batch(bc, ba, eng.make_params(args.experts, H, W, n_local, seed=1305, call=i * Bf, **kw), scores_out=bscores)
            torch.cuda.synchronize()
            tb = time.perf_counter()
            for i in range(nb):
                eng.forward_batch(bc, ba, eng.make_params(args.experts, H, W, n_local, seed=1305, call=(3 + i) * Bf, **kw), scores_out=bscores)
            torch.cuda.synchronize()
            tb = time.perf_counter() - tb
            out["batched"] = {"frames_per_launch": Bf, "launches_timed": nb, "ms_per_batch": tb / nb * 1e3,
                              "value": Bf * n_local * nb / tb, "unit": "hypotheses/s",
                              "note": "esac.forward_batch: frame b == the b-th of B sequential forward calls, bit for bit"}
        if not args.no_training and world == 1:
            # extra figure (not `value`): the training path, esac.backward = esac_hip_backward, same workload
            eng.set_timing(False)
            gts = [np.asarray(f["gt_pose"], np.float32) for f in frames]
            grads = torch.zeros_like(d_coords[0])
            nt = max(5, min(50, args.steps // 8))
            slots = 0
            for i in range(3 + nt):
                if i == 3:
                    torch.cuda.synchronize()
                    tt = time.perf_counter()
                k = i % n_frames
                grads.zero_()
                o = eng.backward_device(d_coords[k], grads, d_assign[k][:n_local], gts[k], 1.0, 100.0, 100.0,
                                        eng.make_params(args.experts, H, W, n_local, seed=1305, call=i, **kw))
                slots += int(o[1]) if i >= 3 else 0
            torch.cuda.synchronize()
            tt = time.perf_counter() - tt
            out["training"] = {"entry": "esac_hip_backward", "ms_per_call": tt / nt * 1e3, "value": n_local * nt / tt,
                               "unit": "hypotheses/s", "refined_hypotheses_per_call": slots / nt,
                               "note": "expected loss + gradient wrt the [E,3,H,W] coordinates, blocking call incl. the zeroing of the gradient tensor"}
            if not args.no_cpu_baseline:
                from oracle import esac_oracle as O
                ts = []
                for i in range(1 + 5):
                    f, ha = frames[i % n_frames], assigns[i % n_frames][:n_local]
                    g_ref = np.zeros_like(f["coords"])
                    t0 = time.time()
                    O.backward(f["coords"], g_ref, ha, gts[i % n_frames], focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"],
                               sub_sampling=f["sub"], seed=1305, call=i)
                    if i >= 1:
                        ts.append(time.time() - t0)
                out["training"]["cpu_oracle_ms_per_call"] = float(np.median(ts)) * 1e3
                out["training"]["cpu_threads"] = O.max_threads()
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(frames, [a[:n_local] for a in assigns], n_local)
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
