#!/usr/bin/env python
"""Markdown tables for DESIGN.md / BASELINE.md from the committed bench lines and kernel profiles of a round.
python scripts/dev/make_tables.py [round, default r05] [dir, default profiles]"""
import json
import os
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
d = sys.argv[2] if len(sys.argv) > 2 else "profiles"


def load(name):
    p = os.path.join(d, "%s_bench_%s.json" % (rnd, name))
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception:
        return None


rows = []
print("| workload | hypotheses/s | ms per call | sample | score | select | refine (µs, live HIP-event stage times) | CPU oracle hyp/s (threads) | × | accuracy vs oracle |")
print("|---|---|---|---|---|---|---|---|---|---|")
for name, label in (("cfg2", "cfg2: 1 expert, 256 hyp, 60×80 (`value`, 400 steps)"), ("cfg2_driver_style", "cfg2, the driver's run (`--steps 20 --warmup 5`)"),
                    ("cfg3", "cfg3: 10 experts, gating, 1024 hyp"), ("cfg4", "cfg4: 12 experts, 4096 hyp, ONE GPU"),
                    ("cfg5a", "cfg5a: 50 experts, Dirichlet(0.3), 16384 hyp, 60×80"), ("cfg5b", "cfg5b: the same, 480×640 maps")):
    b = load(name)
    if not b:
        continue
    st = {k["stage"]: k["avg_us"] for k in b.get("kernels", [])}
    cb = b.get("cpu_baseline") or {}
    acc = b.get("accuracy") or {}
    print("| %s | **%.3f M** | %.4f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s |" % (
        label, b["value"] / 1e6, b["ms_per_step"], st.get("sample", 0), st.get("score", 0), st.get("select_rescore", 0), st.get("refine", 0),
        ("%.1f k (%s, %s)" % (cb["value"] / 1e3, cb.get("cores"), cb.get("kind"))) if cb.get("value") else "—",
        ("%.0f×" % (b["value"] / cb["value"])) if cb.get("value") else "—",
        ("winner %s, median rot err %.1e rad" % (acc.get("winner_match"), acc.get("median_rot_err_rad"))) if acc else "—"))
    extra = []
    if b.get("value_seed1305"):
        extra.append("seed 1305: %.3f M" % (b["value_seed1305"] / 1e6))
    if b.get("value_exact"):
        extra.append("exact routes: %.3f M (%.4f ms)" % (b["value_exact"]["value"] / 1e6, b["value_exact"]["ms_per_step"]))
    if b.get("value_fast"):
        extra.append("fp32 ranking route: %.3f M (%.4f ms)" % (b["value_fast"]["value"] / 1e6, b["value_fast"]["ms_per_step"]))
    if b.get("sharded_world1") and b["sharded_world1"].get("ms_per_step"):
        sw = b["sharded_world1"]
        extra.append("sharded call path at world 1: %.4f ms (+%.1f us over the plain call; all-reduce %.1f us on the GPU, %.1f us of host time in the call)" % (
            sw["ms_per_step"], sw["overhead_us"], (sw.get("allreduce_gpu_ms") or 0) * 1e3, (sw.get("allreduce_host_call_ms") or 0) * 1e3))
    if b.get("batched"):
        extra.append("batched %d frames: %.1f M" % (b["batched"]["frames_per_launch"], b["batched"]["value"] / 1e6))
    if b.get("training"):
        extra.append("backward %.3f ms (CPU %.0f ms)" % (b["training"]["ms_per_call"], b["training"].get("cpu_oracle_ms_per_call", float("nan"))))
    if b.get("with_h2d"):
        extra.append("CPU tensors in: %.3f M" % (b["with_h2d"]["value"] / 1e6))
    if extra:
        rows.append("%s — %s" % (name, "; ".join(extra)))
    rf = b.get("roofline") or {}
    rows.append("%s roofline: bound %s achieved %.4g %s frac %.3f; hbm_nominal %s; traffic %s; valu frac %s; issue-bound frac %s; stale %s" % (
        name, str(rf.get("bound"))[:24], rf.get("achieved", 0), rf.get("unit"), rf.get("frac", 0), (rf.get("hbm_nominal") or {}).get("frac"),
        rf.get("traffic"), (rf.get("valu") or {}).get("frac"), (rf.get("valu") or {}).get("frac_of_issue_bound_rocprofv3"), b.get("profile_stale")))
for b in ("batch16", "batch32", "batch256"):
    x = load(b)
    if x and x.get("batched"):
        rows.append("%s — %d frames per launch: %.1f M hypotheses/s" % (b, x["batched"]["frames_per_launch"], x["batched"]["value"] / 1e6))
print()
for r in rows:
    print("*", r)
print()
print("| kernel (workload) | rocprofv3 avg | issue bound | fraction | VALU busy | fetch MB (×2) | L2 hit |")
print("|---|---|---|---|---|---|---|")
for cfg in ("cfg2", "cfg3", "cfg4", "cfg5a", "cfg5b"):
    p = os.path.join(d, "%s_%s_kernels.json" % (rnd, cfg))
    if not os.path.exists(p):
        continue
    k = json.load(open(p))
    for kk in k["kernels"][:4]:
        print("| `%s` (%s) | %.1f µs | %s | %s | %s | %s | %s |" % (
            kk["name"].replace("void esac::", "").replace("(esac::KArgs)", ""), cfg, kk["avg_us"],
            ("%.1f µs" % kk["issue_bound_us"]) if kk.get("issue_bound_us") is not None else "—",
            ("%.2f" % kk["frac_of_issue_bound"]) if kk.get("frac_of_issue_bound") is not None else "—",
            ("%.2f" % kk["valu_busy_frac"]) if kk.get("valu_busy_frac") is not None else "—",
            ("%.2f" % (kk["fetch_bytes_x2corr"] / 1e6)) if kk.get("fetch_bytes_x2corr") is not None else "—",
            ("%.3f" % kk["l2_hit_rate"]) if kk.get("l2_hit_rate") is not None else "—"))
