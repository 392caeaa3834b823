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
    # 1..4 cells per lane x {winner of a single frame with a team of <= 8 / <= 16 / <= 32, training slots, frames of a small batch}
    assert len(team) == 20
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


def _isa_functions(source, tmp_path):
    """hipcc -S of one source: {mangled kernel name: [instruction lines]}."""
    asm = tmp_path / "k.s"
    out = subprocess.run([HIPCC] + [f for f in B.FLAGS if f not in ("-shared", "-fPIC")] +
                         ["-S", "--cuda-device-only", os.path.join(B.CSRC, source), "-o", str(asm)],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    funcs, cur = {}, None
    for line in asm.read_text().splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        s = line.strip()
        if cur is not None and s and not s.startswith((";", ".")) and not s.endswith(":"):
            cur.append(s)
    return funcs


def test_team_kernel_stays_within_its_instruction_budget(tmp_path):
    """The team refinement is bound by its instruction COUNT (one wavefront per SIMD issues one instruction every ~5.7 cycles
    whatever it is: LAB_NOTES.md, round 4).  Three things that once cost it a fifth of its time and show in the ISA:
    an address materialised in front of every LDS access (arrays laid out behind the 96 KB pad), copies between the two
    register files around every pass (the loop carried the normal equations), tied copies in front of DPP moves."""
    f = _isa_functions("esac_refine_team.hip", tmp_path)
    # the headline call's instantiation since round 6: two cells per lane, a single frame, teams of 9..16 (ten members on the 60x80
    # grid).  One exchange path per instantiation (the width is a template parameter): as run-time branches the three paths took
    # the eight-member kernel from 9,945 to 11,741 instructions and cost it 1.5 us a call (profiles/r06_ab_team10.txt)
    name = [n for n in f if "k_refine_teamILi2ELi0ELi16EE" in n]
    assert len(name) == 1
    ins = f[name[0]]
    assert len(ins) < 10000, len(ins)                                   # 9,144 (three cells per lane, eight members: 9,945; round 4 before the trims: 13,526)
    lds_literals = [i for i in ins if re.match(r"v_mov_b32_e32 v\d+, 0x1[0-9a-f]{4}$", i)]
    assert not lds_literals, lds_literals[:3]                           # LDS addresses fit the offset field
    acc = [i for i in ins if i.startswith("v_accvgpr_")]
    assert len(acc) < 200, len(acc)                                     # 81 (three cells per lane: 189-221; 685 in round 4 before the trims)
    for n, other in f.items():                                          # no instantiation carries another's exchange path
        if "k_refine_team" in n:
            assert len(other) < 11600, (n, len(other))
    dpp = [i for i in ins if i.startswith("v_mov_b32_dpp")]
    tied = [i for i in dpp if re.match(r"v_mov_b32_dpp (v\d+), \1 ", i)]
    assert len(tied) <= len(dpp) // 4, (len(tied), len(dpp))            # quad_perm / mirror moves read their source directly


def _vregs(op):
    """v5 -> {5}; v[12:13] -> {12, 13}; anything else -> empty."""
    m = re.match(r"^-?\|?v(\d+)\|?$", op)
    if m:
        return {int(m.group(1))}
    m = re.match(r"^-?\|?v\[(\d+):(\d+)\]\|?$", op)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def test_lane_dealt_lm_step_is_in_the_isa_and_its_dpp_hazards_are_covered(tmp_path):
    """Round 5: the serial section of an LM round runs on v_fmac_f64_dpp ... row_newbcast (lm_lanes.hpp) -- INLINE ASSEMBLY, which
    the compiler's hazard recogniser does not look into.  gfx9 rule checked here on the generated ISA: a VALU write of a VGPR
    needs 2 wait states before a DPP read of it (an instruction in between is one wait state, s_nop N is N + 1).  Inside a basic
    block the check is exact; every asm block whose DPP sources may be fresh starts with its own s_nop."""
    f = _isa_functions("esac_refine_team.hip", tmp_path)
    for name, ins in f.items():
        if "k_refine_team" not in name:
            continue
        fm = [i for i, t in enumerate(ins) if t.startswith("v_fmac_f64_dpp")]
        assert len(fm) >= 60, (name, len(fm))  # 18 + 18 + 9 of the transform, 30 of the Gauss-Jordan steps
        for i in fm:
            ops = [o.strip() for o in ins[i].split(None, 1)[1].split(",")]
            src = _vregs(ops[1].split()[0])  # the DPP operand (src0)
            assert src, ins[i]
            waited, j = 0, i - 1
            while waited < 2 and j >= 0:
                t = ins[j]
                if t.startswith("s_nop"):
                    waited += int(t.split()[1]) + 1
                else:
                    if t.startswith("v_") and not t.startswith("v_cmp"):
                        dst = _vregs(t.split(None, 1)[1].split(",")[0].strip())
                        assert not (dst & src), (name, ins[j], ins[i])
                    waited += 1
                j -= 1


def test_team_exchange_has_the_local_and_the_written_through_store(tmp_path):
    """Round 5: the granules of a team whose census showed ONE XCD are plain 16-byte stores (they stay in that XCD's L2, where the
    pollers' loads find them); a spread team writes them through (sc1: valid between any two CUs, but visible only when the write
    has reached the home I/O die of its page -- the 67-vs-71 us state of this kernel, profiles/r05_gran_placement_before.txt).
    Both forms must be in every team kernel, the pollers' loads must bypass the L1 (sc1), and no granule may be stored any other way."""
    f = _isa_functions("esac_refine_team.hip", tmp_path)
    teams = {n: ins for n, ins in f.items() if "k_refine_team" in n}
    assert len(teams) == 20
    for name, ins in teams.items():
        st = [i for i in ins if i.startswith("global_store_dwordx4")]
        plain = [i for i in st if not re.search(r"\bsc[01]\b|\bnt\b", i)]
        through = [i for i in st if re.search(r"\bsc1\b", i) and not re.search(r"\bsc0\b", i)]
        assert plain and through, (name, st)
        assert len(plain) + len(through) == len(st), (name, st)
        ld = [i for i in ins if i.startswith("global_load_dwordx4") and re.search(r"\bsc1\b", i)]
        assert ld, name  # gran_load: the polls
