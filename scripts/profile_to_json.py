#!/usr/bin/env python
"""rocprofv3 outputs of one bench.py workload -> profiles/rNN_<config>_kernels.json (+ a text summary).

usage: profile_to_json.py --config cfg2 --round r02 --stats <stats.db> [--pmc <pmc1.db> <pmc2.db> ...]
                          [--command "..."] [--out-dir profiles]

Per kernel: calls, avg / per-step duration (rocprofv3 --kernel-trace --stats), launch shape, and from the PMC passes
(each its own run, as the guide prescribes): FETCH_SIZE (x2: gfx950 tallies the 128-byte requests of wide coalesced
reads at 64 bytes, MI355X_MICROARCH.md "HBM") and WRITE_SIZE in bytes per launch, L2 hit rate, VALU instruction counts,
VALU-busy fraction and an ISSUE BOUND: the time the kernel's own VALU instruction counts need on the SIMDs it occupies at
the measured issue costs (scripts/dev/valu_rate.hip, 4 wavefronts per SIMD: 3.0 cycles per plain fp32/int wave-
instruction, 8.7 per fp32 transcendental, 4.9 per fp64 FMA/ADD/MUL), nominal 2.4 GHz.  bench.py merges this file into
its `kernels` / `roofline` objects; DESIGN.md and BASELINE.md quote it."""
import argparse
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esac_amd import build as _build  # noqa: E402

CYC_PLAIN, CYC_TRANS32, CYC_F64, CLOCK_MHZ, SIMDS = 3.0, 8.7, 4.9, 2400.0, 1024


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--round", default="r02")
    ap.add_argument("--stats", required=True)
    ap.add_argument("--pmc", nargs="*", default=[])
    ap.add_argument("--command", default="")
    ap.add_argument("--out-dir", default="profiles")
    ap.add_argument("--head", default="", help="git commit the profiled tree belongs to (the GPU box has no .git: stamped when the summary is copied into profiles/)")
    a = ap.parse_args()
    db = sqlite3.connect(a.stats)
    kernels = {}
    for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        if "esac::" not in name:
            continue
        kernels[name] = {"name": name, "calls": int(calls), "total_us": total, "avg_us": avg, "pct": pct}
    for r in db.execute("select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), "
                        "max(grid_y), max(workgroup_x) from kernels group by name"):
        if r[0] in kernels:
            kernels[r[0]].update(vgpr=int(r[1]), agpr=int(r[2]), sgpr=int(r[3]), lds_bytes=int(r[4]),
                                 grid=[int(r[5]), int(r[6])], workgroup=int(r[7]))
    counters = {}
    for path in a.pmc:
        d = sqlite3.connect(path)
        for k, c, n, mean in d.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                       "group by kernel_name, counter_name"):
            counters.setdefault(k, {})[c] = mean
    steps = max([k["calls"] for n, k in kernels.items() if "k_refine" in n] + [1])
    for name, k in kernels.items():
        k["per_step_us"] = k["total_us"] / steps
        c = counters.get(name, {})
        if "FETCH_SIZE" in c:
            k["fetch_bytes_x2corr"] = 2.0 * 1024.0 * c["FETCH_SIZE"]
        if "WRITE_SIZE" in c:
            k["write_bytes"] = 1024.0 * c["WRITE_SIZE"]
        if "TCC_HIT_sum" in c and c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0) > 0:
            k["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        if "SQ_INSTS_VALU" in c:
            valu = c["SQ_INSTS_VALU"]
            trans = c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
            f64 = sum(c.get(x, 0.0) for x in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64"))
            k["valu_insts"], k["valu_trans_insts"], k["valu_f64_insts"] = valu, trans, f64
            wg = k.get("workgroup", 64)
            n_wg = (k.get("grid", [wg, 1])[0] // max(wg, 1)) * max(1, k.get("grid", [wg, 1])[1])
            waves = n_wg * max(1, wg // 64)
            if "k_refine_team" in name:  # every eighth workgroup of the launch is a member, the others leave at once (placement)
                waves = max(1, n_wg // 8) * max(1, wg // 64)
            simds = min(SIMDS, max(1, waves))
            k["simds_occupied"] = simds
            plain = max(0.0, valu - trans - f64)
            cycles = plain * CYC_PLAIN + trans * CYC_TRANS32 + f64 * CYC_F64
            k["issue_bound_us"] = cycles / simds / CLOCK_MHZ
            k["frac_of_issue_bound"] = k["issue_bound_us"] / k["avg_us"] if k["avg_us"] > 0 else None
            if "SQ_ACTIVE_INST_VALU" in c:  # quad-cycles summed over SIMDs
                k["valu_busy_frac"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (k["avg_us"] * CLOCK_MHZ * simds)
        k["counters"] = {n: v for n, v in c.items()}
    out = {"config": a.config, "round": a.round, "command": a.command, "steps_profiled": steps,
           # what was measured: hash of esac_amd/csrc + include/esac_hip.h (bench.py flags the profile as stale when the
           # running tree differs) and the commit it belongs to
           "csrc_sha16": _build.source_hash(), "head": a.head,
           "issue_cost_model": {"cycles_per_wave_instruction": {"plain_valu": CYC_PLAIN, "fp32_transcendental": CYC_TRANS32, "fp64": CYC_F64},
                                "clock_mhz": CLOCK_MHZ, "source": "scripts/dev/valu_rate.hip on MI355X (profiles/%s_valu_rate.txt)" % a.round},
           "kernels": sorted(kernels.values(), key=lambda k: -k["total_us"])}
    os.makedirs(a.out_dir, exist_ok=True)
    base = os.path.join(a.out_dir, "%s_%s_kernels" % (a.round, a.config))
    with open(base + ".json", "w") as fh:
        json.dump(out, fh, indent=1)
    with open(base + ".txt", "w") as fh:
        fh.write("# %s %s -- %s\n# rocprofv3 --kernel-trace --stats + PMC passes (separate runs); %d forward calls traced\n" % (a.round, a.config, a.command, steps))
        fh.write("%-48s %6s %10s %10s %6s %5s %7s %12s %10s %7s %10s %9s %8s\n" % (
            "kernel", "calls", "avg_us", "per_step", "pct", "vgpr", "lds_B", "fetch_MB_x2", "write_MB", "L2hit", "issue_bnd", "of_bound", "valubusy"))
        for k in out["kernels"]:
            g = lambda key, scale=1.0, fmt="%.2f": (fmt % (k[key] * scale)) if k.get(key) is not None else "-"
            fh.write("%-48s %6d %10.3f %10.3f %5.1f%% %5s %7s %12s %10s %7s %10s %9s %8s\n" % (
                k["name"].replace("void ", "").replace("(esac::KArgs)", "")[:48], k["calls"], k["avg_us"], k["per_step_us"], k["pct"],
                k.get("vgpr", "-"), k.get("lds_bytes", "-"), g("fetch_bytes_x2corr", 1e-6, "%.3f"), g("write_bytes", 1e-6, "%.3f"),
                g("l2_hit_rate", 1.0, "%.3f"), g("issue_bound_us", 1.0, "%.2f"), g("frac_of_issue_bound", 1.0, "%.2f"), g("valu_busy_frac", 1.0, "%.2f")))
    print(open(base + ".txt").read())


if __name__ == "__main__":
    main()
