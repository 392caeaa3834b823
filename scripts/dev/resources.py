#!/usr/bin/env python
"""Compact table of the kernels' compile-time resources (hipcc -Rpass-analysis=kernel-resource-usage, no GPU needed).
usage: python scripts/dev/resources.py [source.hip ...]   (default: every source of the library)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from esac_amd import build as B  # noqa: E402


def usage(source, extra=()):
    with tempfile.TemporaryDirectory() as td:
        out = subprocess.run(["/opt/rocm/bin/hipcc"] + [f for f in B.FLAGS if f not in ("-shared", "-fPIC")] + list(extra) +
                             ["-c", os.path.join(B.CSRC, source), "-o", os.path.join(td, "o.o"),
                              "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    rows, cur = [], None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return rows


if __name__ == "__main__":
    srcs = [a for a in sys.argv[1:] if a.endswith(".hip")] or B.SOURCES
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    for src in srcs:
        for r in usage(src, extra):
            name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
            print("%-58s vgpr %3d agpr %3d sgpr %3d scratch %4d lds %6d occ %d" % (
                name[:58], r.get("VGPRs", -1), r.get("AGPRs", 0), r.get("TotalSGPRs", -1), r.get("ScratchSize", -1),
                r.get("LDS Size", -1), r.get("Occupancy", -1)))
