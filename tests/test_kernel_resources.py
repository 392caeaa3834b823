"""Compile-time guard on the hot kernels' resource usage (hipcc -Rpass-analysis=kernel-resource-usage, no GPU needed):
a change that makes the refinement spill to scratch, or bloats the streaming score kernel's registers, shows up here
before it shows up as a slower bench line."""
import os
import re
import shutil
import subprocess

import pytest

from esac_amd import build as B

HIPCC = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
pytestmark = pytest.mark.skipif(HIPCC is None, reason="hipcc not available")


def _usage(source, tmp_path):
    out = subprocess.run([HIPCC] + [f for f in B.FLAGS if f not in ("-shared", "-fPIC")] +
                         ["-c", os.path.join(B.CSRC, source), "-o", str(tmp_path / "o.o"), "-Rpass-analysis=kernel-resource-usage"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    return kernels


def test_refinement_kernels_do_not_spill(tmp_path):
    k = _usage("esac_refine.hip", tmp_path)
    refine = {n: v for n, v in k.items() if "k_refine" in n}
    # {LDS, global list} x {vector, scalar error pass} x {winner, slots} + the cooperating-workgroups winner variant of the
    # large grids (the team of the small grids is a kernel of its own, below)
    assert len(refine) == 9
    for name, u in refine.items():
        assert u["ScratchSize"] == 0, (name, u)
        # one wavefront per SIMD by design: the pose state, 24 accumulators and two correspondences in flight
        assert u["VGPRs"] + u.get("AGPRs", 0) <= 512, (name, u)
        if "ELb0E" in name.split("k_refineILi256")[1][:6]:  # LDS-list variants hold the 128 KiB list + reduction scratch
            assert 128 * 1024 <= u["LDS"] <= 160 * 1024, (name, u)


def test_team_refinement_kernels_do_not_spill(tmp_path):
    k = _usage("esac_refine_team.hip", tmp_path)
    team = {n: v for n, v in k.items() if "k_refine_team" in n}
    assert len(team) == 8  # 1..4 cells per lane x {winner of a single frame, training slots}
    for name, u in team.items():
        assert u["ScratchSize"] == 0, (name, u)
        assert u["VGPRs"] + u.get("AGPRs", 0) <= 512, (name, u)
        assert 80 * 1024 < u["LDS"] <= 160 * 1024, (name, u)  # more than half a CU's LDS: one member per CU


def test_streaming_and_selection_kernels_stay_lean(tmp_path):
    k = _usage("esac_kernels.hip", tmp_path)
    score = {n: v for n, v in k.items() if "k_score_fast" in n}
    assert score
    for name, u in score.items():
        assert u["ScratchSize"] == 0 and u["VGPRs"] <= 72, (name, u)   # 7 wavefronts per SIMD (the packed-fp32 form holds the pose as register pairs)
    sel = [v for n, v in k.items() if "k_select_rescore" in n]
    assert sel and all(u["ScratchSize"] == 0 and u["VGPRs"] <= 128 for u in sel)  # 1024-thread workgroups
    quad = [v for n, v in k.items() if "k_sampleILi256ELi2" in n]
    assert quad and quad[0]["ScratchSize"] == 0                         # the single-frame sampler runs from registers
    pre = [v for n, v in k.items() if "k_sample_prescreen" in n]
    assert pre and pre[0]["ScratchSize"] == 0 and pre[0]["VGPRs"] <= 256        # the fp64 screen: 2 wavefronts per SIMD
