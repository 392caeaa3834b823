"""In-tree build of the HIP extension (libesac_hip.so) for gfx950.

`hipcc --offload-arch=gfx950` cross-compiles without a GPU.  The shared object
is written next to this file so that it travels with the source tree (it is
git-ignored, not gpurun-ignored).
"""
import glob
import hashlib
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libesac_hip.so")
SOURCES = ["esac_kernels.hip", "esac_score_tiled.hip", "esac_refine.hip", "esac_refine_team.hip", "esac_backward.hip", "esac_capi.hip"]
# every header under csrc/ (a new one must not be forgotten here: a stale library would be tested against new headers)
HEADERS = sorted(os.path.basename(h) for h in glob.glob(os.path.join(CSRC, "*.hpp"))) + [os.path.join("..", "..", "include", "esac_hip.h")]
# -ffp-contract=off: the fp64 "exact" kernels follow IEEE op-by-op like the CPU
# code they are compared with; the fp32 streaming kernel asks for FMAs explicitly.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]
# RCCL (the multi-GPU score exchange, esac_hip_allreduce_sum) is bound by dlopen at the first esac_hip_comm_* call, not linked
LINK = ["-ldl"]


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources, headers and the C ABI header: what a profile under profiles/
    was measured on (scripts/profile_to_json.py stamps it, bench.py compares it with the running tree)."""
    h = hashlib.sha256()
    for name in sorted(SOURCES + HEADERS):
        with open(os.path.join(CSRC, name), "rb") as fh:
            h.update(os.path.basename(name).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libesac_hip.so")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


OBJ_DIR = os.path.join(_HERE, "build")  # git-ignored; objects are keyed by (source, flags, contents of every dependency)


def _compile_and_link(out_path, extra_flags=(), force=False, verbose=False):
    """One translation unit per hipcc process, side by side (the six sources are independent: a minute becomes the longest file's
    ~25 s), objects reused when neither the source, a header nor the flags changed; then one link."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    cflags = [f for f in FLAGS if f != "-shared"] + list(extra_flags)
    dep = hashlib.sha256()
    for name in sorted(HEADERS):
        with open(os.path.join(CSRC, name), "rb") as fh:
            dep.update(fh.read())
    dep.update(" ".join(cflags).encode())

    def one(src):
        with open(os.path.join(CSRC, src), "rb") as fh:
            key = hashlib.sha256(dep.digest() + fh.read()).hexdigest()[:16]
        obj = os.path.join(OBJ_DIR, "%s.%s.o" % (os.path.splitext(src)[0], key))
        if force or not os.path.exists(obj):
            cmd = [_hipcc()] + cflags + ["-c", os.path.join(CSRC, src), "-o", obj + ".tmp"]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(obj + ".tmp", obj)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(one, SOURCES))
    keep = set(objs)
    for old in glob.glob(os.path.join(OBJ_DIR, "*.o")):  # objects of older revisions of the default build
        if old not in keep and not extra_flags and os.path.getmtime(old) < min(os.path.getmtime(o) for o in objs) - 86400:
            os.remove(old)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + LINK + ["-o", out_path + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(out_path + ".tmp", out_path)
    return out_path


def build_hip(force=False, verbose=False):
    if not force and not needs_build():
        return LIB_PATH
    return _compile_and_link(LIB_PATH, force=force, verbose=verbose)


def build_variant(out_path, extra_flags=()):
    """The same sources with extra compiler flags into another file (measurement scripts: ESAC_HIP_LIB=<out_path>)."""
    return _compile_and_link(out_path, extra_flags=extra_flags)


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1:  # python esac_amd/build.py <out.so> [-DFLAG ...]
        print(build_variant(sys.argv[1], sys.argv[2:]))
    else:
        print(build_hip(force=True, verbose=True))
