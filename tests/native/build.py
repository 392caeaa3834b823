"""Builds the TEST-ONLY host probe of the device math headers (pose_math.hpp / lm_math.hpp compiled for the
host with hipcc) so that the CPU suite can exercise the kernels' own source against the oracle."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_math_probe.cpp")
LIB = os.path.join(HERE, "libhost_math_probe.so")
CSRC = os.path.join(HERE, "..", "..", "esac_amd", "csrc")


def build(force=False):
    deps = [SRC] + [os.path.join(CSRC, h) for h in ("pose_math.hpp", "lm_math.hpp", "bwd_math.hpp", "lm_lanes.hpp")]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC",
                               "-shared", SRC, "-o", LIB])
    return LIB


def build_abi_check():
    """gcc -std=c99 build of abi_check.c: the header must be plain C, the library a plain C ABI."""
    src = os.path.join(HERE, "abi_check.c")
    exe = os.path.join(HERE, "abi_check")
    hdr = os.path.join(HERE, "..", "..", "include", "esac_hip.h")
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in (src, hdr)):
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O1", src, "-o", exe, "-ldl"])
    return exe


SCREEN_SRC = os.path.join(HERE, "p3p_screen_probe.cpp")
SCREEN_LIB = os.path.join(HERE, "libp3p_screen_probe.so")


def build_screen_probe(force=False):
    """Host build (OpenMP) of the fp32 sampling screen + the fp64 route it screens for."""
    deps = [SCREEN_SRC] + [os.path.join(CSRC, h) for h in ("pose_math.hpp", "p3p_screen.hpp")]
    if force or not os.path.exists(SCREEN_LIB) or any(os.path.getmtime(d) > os.path.getmtime(SCREEN_LIB) for d in deps):
        hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                               "-fopenmp", "-include", "omp.h", "-Wno-unused-result", SCREEN_SRC, "-o", SCREEN_LIB])
    return SCREEN_LIB


CAMPAIGN_SRC = os.path.join(HERE, "screen_campaign.hip")
CAMPAIGN_LIB = os.path.join(HERE, "libscreen_campaign.so")


def build_screen_campaign(force=False):
    """DEVICE build of the screen-vs-exact-route campaign (screen_campaign.hip): the product's own compiler flags, so the
    screen runs the arithmetic the kernels run (v_rcp / v_rsq / v_sqrt estimates, the same contraction)."""
    from esac_amd import build as product
    deps = [CAMPAIGN_SRC] + [os.path.join(CSRC, h) for h in ("pose_math.hpp", "p3p_screen.hpp")]
    if force or not os.path.exists(CAMPAIGN_LIB) or any(os.path.getmtime(d) > os.path.getmtime(CAMPAIGN_LIB) for d in deps):
        hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
        subprocess.check_call([hipcc] + product.FLAGS + [CAMPAIGN_SRC, "-o", CAMPAIGN_LIB])
    return CAMPAIGN_LIB


FILLER_SRC = os.path.join(HERE, "filler.hip")
FILLER_LIB = os.path.join(HERE, "libfiller.so")


def build_filler(force=False):
    """DEVICE build of the CU filler (filler.hip): workgroups that hold the CUs of one XCD and leave one by one."""
    if force or not os.path.exists(FILLER_LIB) or os.path.getmtime(FILLER_SRC) > os.path.getmtime(FILLER_LIB):
        hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result", FILLER_SRC, "-o", FILLER_LIB])
    return FILLER_LIB
