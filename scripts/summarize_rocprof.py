#!/usr/bin/env python
"""Turn rocprofv3 rocpd (.db) outputs into the text summaries committed under profiles/.

usage: summarize_rocprof.py <stats.db> [<pmc_fetch.db> <pmc_write.db>]
FETCH_SIZE on gfx950 reports 1/2 of the bytes of wide coalesced streaming reads
(MI355X_MICROARCH.md, HBM section) -- the corrected column doubles it; WRITE_SIZE is uncalibrated.
"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    print("== rocprofv3 --kernel-trace --stats (durations in us) ==")
    print("%-52s %8s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in db.execute(
            "select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-52s %8d %12.3f %10.3f %6.2f%%" % (name[:52], calls, total, avg, pct))
    print()
    print("%-52s %6s %6s %6s %8s  %s" % ("kernel", "vgpr", "agpr", "sgpr", "lds_B", "grid x wg"))
    for r in db.execute("select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), "
                        "max(grid_x), max(workgroup_x) from kernels group by name"):
        print("%-52s %6d %6d %6d %8d  %d x %d" % (r[0][:52], r[1], r[2], r[3], r[4], r[5], r[6]))
    if len(sys.argv) > 2:
        print()
        print("== PMC (separate passes: --pmc FETCH_SIZE / --pmc WRITE_SIZE), KB per launch ==")
        vals = {}
        for path in sys.argv[2:]:
            d = sqlite3.connect(path)
            for k, c, n, mean in d.execute("select kernel_name, counter_name, count(*), avg(value) "
                                           "from counters_collection group by kernel_name, counter_name"):
                vals.setdefault(k, {})[c] = (n, mean)
        print("%-52s %10s %14s %10s" % ("kernel", "FETCH_KB", "FETCH_KB_x2corr", "WRITE_KB"))
        for k, v in sorted(vals.items()):
            f = v.get("FETCH_SIZE", (0, float("nan")))[1]
            w = v.get("WRITE_SIZE", (0, float("nan")))[1]
            print("%-52s %10.2f %14.2f %10.2f" % (k[:52], f, 2 * f, w))


if __name__ == "__main__":
    main()
