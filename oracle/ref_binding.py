"""ctypes binding of oracle/_ref/libesac_ref.so -- the reference's own esac_util.h / esac_types.h /
thread_rand.cpp (compiled from /root/reference against oracle/ref_shim) driven in esac_forward's order.
TEST INFRASTRUCTURE ONLY.  Built by `make -C oracle ref` where /root/reference exists; the .so travels to the GPU box."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libesac_ref.so")
REFERENCE_DIR = "/root/reference/code/esac"


def build(force=False):
    """Compile oracle/_ref from the reference sources where they lie (only possible where they are mounted)."""
    if not os.path.exists(os.path.join(REFERENCE_DIR, "esac_util.h")):
        return LIB_PATH if os.path.exists(LIB_PATH) else None
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(LIB_PATH)
        os.environ.setdefault("OMP_NUM_THREADS", "1")
        _lib = C.CDLL(LIB_PATH)
        vp, i, f, u = C.c_void_p, C.c_int, C.c_float, C.c_uint
        _lib.ref_forward.argtypes = [vp, i, i, i, vp, i, vp, i, i, f, f, f, f, f, f, f, i, u, u, vp, vp, vp, vp, vp, vp, vp]
        _lib.ref_forward.restype = i
        _lib.ref_esac_forward.argtypes = [vp, i, i, i, vp, i, vp, i, i, f, f, f, f, f, f, f, i]
        _lib.ref_esac_forward.restype = i
        _lib.ref_esac_backward.argtypes = [vp, vp, i, i, i, vp, i, vp, f, f, f, i, i, f, f, f, f, f, f, f, i]
        _lib.ref_esac_backward.restype = C.c_double
        _lib.ref_rng_reset.argtypes = [u]
        _lib.ref_replay_irand.argtypes = [i, i, vp]
        _lib.ref_replay_irand.restype = i
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def forward(coords, assign, shift_x=0, shift_y=0, focal=525.0, ppx=320.0, ppy=240.0, inlier_thresh=10.0,
            inlier_alpha=100.0, inlier_beta=0.5, max_reproj=100.0, sub_sampling=8, seed=1305, max_tries=1000000,
            max_ref_steps=100):
    """Run the reference's code (single OpenMP thread, its own mt19937 stream re-seeded with `seed`)."""
    import threadpoolctl  # the reference's omp pragmas must run on ONE thread for a defined RNG order
    sc = np.ascontiguousarray(coords, np.float32)
    ha = np.ascontiguousarray(assign, np.int64)
    E, _, H, W = sc.shape
    N = len(ha)
    out = dict(pose=np.zeros((4, 4), np.float32), sample_xy=np.zeros((N, 4, 2), np.int32), hyps=np.zeros((N, 6)),
               scores=np.zeros(N), winner=np.zeros(1, np.int32), refined=np.zeros(6), inlier_map=np.zeros((H, W), np.uint8),
               entropy=np.zeros(1))
    L = lib()
    with threadpoolctl.threadpool_limits(limits=1, user_api="openmp"):
        L.ref_rng_reset(int(seed))
        e = L.ref_forward(_p(sc), E, H, W, _p(ha), N, _p(out["pose"]), int(shift_x), int(shift_y), focal, ppx, ppy,
                          inlier_thresh, inlier_alpha, inlier_beta, max_reproj, int(sub_sampling), int(max_tries),
                          int(max_ref_steps), _p(out["sample_xy"]), _p(out["hyps"]), _p(out["scores"]), _p(out["winner"]),
                          _p(out["refined"]), _p(out["inlier_map"]), _p(out["entropy"]))
    out["expert"] = e
    out["winner"] = int(out["winner"][0])
    out["entropy"] = float(out["entropy"][0])
    return out


def esac_forward(coords, assign, shift_x=0, shift_y=0, focal=525.0, ppx=320.0, ppy=240.0, inlier_thresh=10.0,
                 inlier_alpha=100.0, inlier_beta=0.5, max_reproj=100.0, sub_sampling=8, seed=1305):
    """The reference's esac_forward (esac.cpp:64-190) itself, called the way pybind11 would call it."""
    import threadpoolctl
    sc = np.ascontiguousarray(coords, np.float32)
    ha = np.ascontiguousarray(assign, np.int64)
    E, _, H, W = sc.shape
    pose = np.zeros((4, 4), np.float32)
    L = lib()
    with threadpoolctl.threadpool_limits(limits=1, user_api="openmp"):
        L.ref_rng_reset(int(seed))
        e = L.ref_esac_forward(_p(sc), E, H, W, _p(ha), len(ha), _p(pose), int(shift_x), int(shift_y), focal, ppx, ppy,
                               inlier_thresh, inlier_alpha, inlier_beta, max_reproj, int(sub_sampling))
    return e, pose


def esac_backward(coords, out_gradients, assign, gt_pose, w_rot=1.0, w_trans=100.0, loss_cut=100.0, shift_x=0,
                  shift_y=0, focal=525.0, ppx=320.0, ppy=240.0, inlier_thresh=10.0, inlier_alpha=100.0, inlier_beta=0.5,
                  max_reproj=100.0, sub_sampling=8, seed=1305):
    """The reference's esac_backward (esac.cpp:213-511) itself: accumulates into out_gradients, returns the loss."""
    import threadpoolctl
    sc = np.ascontiguousarray(coords, np.float32)
    ha = np.ascontiguousarray(assign, np.int64)
    gt = np.ascontiguousarray(gt_pose, np.float32).reshape(4, 4)
    assert out_gradients.dtype == np.float32 and out_gradients.flags.c_contiguous and out_gradients.shape == sc.shape
    E, _, H, W = sc.shape
    L = lib()
    with threadpoolctl.threadpool_limits(limits=1, user_api="openmp"):
        L.ref_rng_reset(int(seed))
        return L.ref_esac_backward(_p(sc), _p(out_gradients), E, H, W, _p(ha), len(ha), _p(gt), w_rot, w_trans, loss_cut,
                                   int(shift_x), int(shift_y), focal, ppx, ppy, inlier_thresh, inlier_alpha, inlier_beta,
                                   max_reproj, int(sub_sampling))


def replay_irand(seed):
    """A Python callable that replays the reference's single-thread mt19937 stream for the oracle's callback RNG."""
    L = lib()
    L.ref_rng_reset(int(seed))
    return lambda lo, hi: L.ref_replay_irand(lo, hi, None)
