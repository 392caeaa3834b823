"""ctypes binding for the CPU oracle (oracle/esac_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py -- never by esac_amd/ (the product).
PARITY UNPINNED: see oracle/esac_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libesac_oracle.so")

IRAND_FN = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, C.c_void_p)


class _Args(C.Structure):
    _fields_ = [
        ("scene_coords", C.c_void_p),
        ("sc_stride", C.c_int64 * 4),
        ("E", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("hyp_assign", C.c_void_p),
        ("assign_stride", C.c_int64),
        ("N", C.c_int),
        ("shift_x", C.c_int), ("shift_y", C.c_int),
        ("focal", C.c_float), ("ppx", C.c_float), ("ppy", C.c_float),
        ("inlier_thresh", C.c_float), ("inlier_alpha", C.c_float),
        ("inlier_beta", C.c_float), ("max_reproj", C.c_float),
        ("sub_sampling", C.c_int),
        ("seed", C.c_uint64), ("call", C.c_uint64),
        ("rng_mode", C.c_int),
        ("irand_cb", IRAND_FN),
        ("irand_user", C.c_void_p),
        ("max_tries", C.c_int), ("max_ref_steps", C.c_int), ("num_threads", C.c_int),
        ("out_pose", C.c_void_p), ("out_sample_xy", C.c_void_p), ("out_tries", C.c_void_p),
        ("out_hyps", C.c_void_p), ("out_scores", C.c_void_p), ("out_probs", C.c_void_p),
        ("out_entropy", C.c_void_p), ("out_winner", C.c_void_p), ("out_refined", C.c_void_p),
        ("out_ref_steps", C.c_void_p), ("out_inlier_counts", C.c_void_p),
        ("out_inlier_map", C.c_void_p), ("out_winner_errs", C.c_void_p),
        ("out_phase_ms", C.c_void_p),
        ("out_lm_iters", C.c_void_p),
        ("hyp_index", C.c_void_p),
        ("in_hyps", C.c_void_p),
    ]


def build(force=False):
    """Compile the oracle's C restatement (gcc, seconds). Building the checker is not using it."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(os.path.join(_HERE, f)) for f in ("esac_oracle.c", "esac_oracle.h", "esac_oracle_bwd.inc")
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.esac_oracle_forward.argtypes = [C.POINTER(_Args)]
        _lib.esac_oracle_forward.restype = C.c_int
        _lib.esac_oracle_max_threads.restype = C.c_int
        d = C.c_double
        _lib.esac_oracle_p3p.argtypes = [C.c_void_p, C.c_void_p, d, d, d, d, C.c_void_p, C.c_void_p]
        _lib.esac_oracle_p3p.restype = C.c_int
        _lib.esac_oracle_p3p_all.argtypes = [C.c_void_p, C.c_void_p, d, d, d, d, C.c_void_p, C.c_void_p]
        _lib.esac_oracle_p3p_all.restype = C.c_int
        _lib.esac_oracle_solve_deg4.argtypes = [d, d, d, d, d, C.c_void_p]
        _lib.esac_oracle_solve_deg4.restype = C.c_int
        _lib.esac_oracle_lm_pnp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, d, d, d, d, C.c_void_p]
        _lib.esac_oracle_lm_pnp.restype = C.c_int
        _lib.esac_oracle_project.argtypes = [C.c_void_p, C.c_void_p, d, d, d, d, C.c_void_p, C.c_int, C.c_void_p]
        _lib.esac_oracle_draw_cells.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
        f = C.c_float
        _lib.esac_oracle_pose_loss.argtypes = [C.c_void_p, C.c_void_p, d, d, d]
        _lib.esac_oracle_pose_loss.restype = d
        _lib.esac_oracle_pose_dloss.argtypes = [C.c_void_p, C.c_void_p, d, d, d, C.c_void_p]
        _lib.esac_oracle_trans2pose.argtypes = [C.c_void_p, C.c_void_p]
        _lib.esac_oracle_dproject_dobj.argtypes = [f, f, f, f, f, C.c_void_p, C.c_void_p, f, f, f, f, C.c_void_p]
        _lib.esac_oracle_norm_jac_row.argtypes = [C.c_void_p, C.c_void_p, f, f, f, f, f, f, f, f, f, C.c_void_p]
        _lib.esac_oracle_norm_jac_row.restype = C.c_int
        _lib.esac_oracle_pinv_sym6.argtypes = [C.c_void_p, C.c_void_p]
    return _lib


def max_threads():
    return lib().esac_oracle_max_threads()


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def forward(scene_coords, hyp_assign, shift_x=0, shift_y=0, focal=525.0, ppx=320.0, ppy=240.0,
            inlier_thresh=10.0, inlier_alpha=100.0, inlier_beta=0.5, max_reproj=100.0, sub_sampling=8,
            seed=1305, call=0, max_tries=0, max_ref_steps=-1, num_threads=0, irand=None, hyp_index=None, in_hyps=None):
    """Run the oracle's esac_forward restatement; returns a dict of every stage output.
    in_hyps: optional float64 [N,6] hypotheses used instead of sampling (scoring, selection, refinement unchanged).

    scene_coords: float32 ndarray [E,3,H,W] (any strides); hyp_assign: int64 ndarray [N] (any stride, 0 ok).
    """
    sc = np.asarray(scene_coords)
    ha = np.asarray(hyp_assign)
    assert sc.dtype == np.float32 and sc.ndim == 4 and sc.shape[1] == 3
    assert ha.dtype == np.int64 and ha.ndim == 1
    E, _, H, W = sc.shape
    N = ha.shape[0]
    nref = 100 if max_ref_steps < 0 else max_ref_steps
    out = dict(
        pose=np.zeros((4, 4), np.float32), sample_xy=np.zeros((N, 4, 2), np.int32),
        tries=np.zeros(N, np.int32), hyps=np.zeros((N, 6), np.float64), scores=np.zeros(N, np.float64),
        probs=np.zeros(N, np.float64), entropy=np.zeros(1, np.float64), winner=np.zeros(1, np.int32),
        refined=np.zeros(6, np.float64), ref_steps=np.zeros(1, np.int32),
        inlier_counts=np.zeros(nref + 1, np.int32), inlier_map=np.zeros((H, W), np.uint8),
        winner_errs=np.zeros((H, W), np.float32), phase_ms=np.zeros(4, np.float64),
        lm_iters=np.zeros(1, np.int32),
    )
    a = _Args()
    a.scene_coords = _p(sc)
    for i in range(4):
        a.sc_stride[i] = sc.strides[i] // 4
    a.E, a.H, a.W = E, H, W
    a.hyp_assign = _p(ha)
    a.assign_stride = ha.strides[0] // 8
    a.N = N
    a.shift_x, a.shift_y = int(shift_x), int(shift_y)
    a.focal, a.ppx, a.ppy = float(focal), float(ppx), float(ppy)
    a.inlier_thresh, a.inlier_alpha = float(inlier_thresh), float(inlier_alpha)
    a.inlier_beta, a.max_reproj = float(inlier_beta), float(max_reproj)
    a.sub_sampling = int(sub_sampling)
    a.seed, a.call = int(seed), int(call)
    cb = None
    if irand is not None:
        cb = IRAND_FN(lambda lo, hi, _u: int(irand(lo, hi)))
        a.rng_mode = 1
        a.irand_cb = cb
    else:
        a.rng_mode = 0
    a.max_tries, a.max_ref_steps, a.num_threads = int(max_tries), int(max_ref_steps), int(num_threads)
    hi = None
    if hyp_index is not None:
        hi = np.ascontiguousarray(hyp_index, np.int32)
        assert hi.shape == (N,)
        a.hyp_index = _p(hi)
    ih = None
    if in_hyps is not None:
        ih = np.ascontiguousarray(in_hyps, np.float64)
        assert ih.shape == (N, 6)
        a.in_hyps = _p(ih)
    for k in ("pose", "sample_xy", "tries", "hyps", "scores", "probs", "entropy", "winner", "refined",
              "ref_steps", "inlier_counts", "inlier_map", "winner_errs", "phase_ms", "lm_iters"):
        setattr(a, "out_" + k, _p(out[k]))
    rc = lib().esac_oracle_forward(C.byref(a))
    if rc < 0:
        raise RuntimeError("esac_oracle_forward failed with code %d" % rc)
    out["expert"] = rc
    out["winner"] = int(out["winner"][0])
    out["entropy"] = float(out["entropy"][0])
    out["ref_steps"] = int(out["ref_steps"][0])
    out["lm_iters"] = int(out["lm_iters"][0])
    return out


class _BwdArgs(C.Structure):
    _fields_ = [
        ("fwd", _Args),
        ("gt_pose", C.c_void_p),
        ("w_rot", C.c_float), ("w_trans", C.c_float), ("loss_cut", C.c_float),
        ("out_gradients", C.c_void_p),
        ("grad_stride", C.c_int64 * 4),
        ("out_probs", C.c_void_p), ("out_losses", C.c_void_p), ("out_init_hyps", C.c_void_p),
        ("out_ref_hyps", C.c_void_p), ("out_score_grads", C.c_void_p), ("out_dloss", C.c_void_p),
        ("out_sample_xy", C.c_void_p), ("out_entropy", C.c_void_p), ("out_grad_path1", C.c_void_p),
        ("out_grad_path2", C.c_void_p),
    ]


def backward(scene_coords, out_gradients, hyp_assign, gt_pose, w_rot=1.0, w_trans=100.0, loss_cut=100.0,
             shift_x=0, shift_y=0, focal=525.0, ppx=320.0, ppy=240.0, inlier_thresh=10.0, inlier_alpha=100.0,
             inlier_beta=0.5, max_reproj=100.0, sub_sampling=8, seed=1305, call=0, max_tries=0, max_ref_steps=-1,
             num_threads=0, irand=None, hyp_index=None, want_paths=False):
    """Oracle restatement of esac.backward (esac.cpp:213-230): accumulates into `out_gradients` (float32 ndarray
    [E,3,H,W], +=) and returns a dict with the expected loss and the stage outputs."""
    sc = np.asarray(scene_coords)
    ha = np.asarray(hyp_assign)
    og = out_gradients
    gt = np.ascontiguousarray(gt_pose, np.float32).reshape(16)
    assert sc.dtype == np.float32 and sc.ndim == 4 and og.dtype == np.float32 and og.shape == sc.shape
    assert ha.dtype == np.int64 and ha.ndim == 1
    E, _, H, W = sc.shape
    N = ha.shape[0]
    out = dict(probs=np.zeros(N), losses=np.zeros(N), init_hyps=np.zeros((N, 6)), ref_hyps=np.zeros((N, 6)),
               score_grads=np.zeros(N), dloss=np.zeros((N, 6)), sample_xy=np.zeros((N, 4, 2), np.int32),
               entropy=np.zeros(1))
    if want_paths:
        out["grad_path1"] = np.zeros((N, H * W, 3))
        out["grad_path2"] = np.zeros((N, H * W, 3))
    b = _BwdArgs()
    a = b.fwd
    a.scene_coords = _p(sc)
    for i in range(4):
        a.sc_stride[i] = sc.strides[i] // 4
        b.grad_stride[i] = og.strides[i] // 4
    a.E, a.H, a.W = E, H, W
    a.hyp_assign = _p(ha)
    a.assign_stride = ha.strides[0] // 8
    a.N = N
    a.shift_x, a.shift_y = int(shift_x), int(shift_y)
    a.focal, a.ppx, a.ppy = float(focal), float(ppx), float(ppy)
    a.inlier_thresh, a.inlier_alpha = float(inlier_thresh), float(inlier_alpha)
    a.inlier_beta, a.max_reproj = float(inlier_beta), float(max_reproj)
    a.sub_sampling = int(sub_sampling)
    a.seed, a.call = int(seed), int(call)
    cb = None
    if irand is not None:
        cb = IRAND_FN(lambda lo, hi, _u: int(irand(lo, hi)))
        a.rng_mode = 1
        a.irand_cb = cb
    a.max_tries, a.max_ref_steps, a.num_threads = int(max_tries), int(max_ref_steps), int(num_threads)
    hi = None
    if hyp_index is not None:
        hi = np.ascontiguousarray(hyp_index, np.int32)
        a.hyp_index = _p(hi)
    b.gt_pose = _p(gt)
    b.w_rot, b.w_trans, b.loss_cut = float(w_rot), float(w_trans), float(loss_cut)
    b.out_gradients = _p(og)
    for k in out:
        setattr(b, "out_" + k, _p(out[k]))
    L = lib()
    L.esac_oracle_backward.argtypes = [C.POINTER(_BwdArgs)]
    L.esac_oracle_backward.restype = C.c_double
    loss = L.esac_oracle_backward(C.byref(b))
    if loss < 0:
        raise RuntimeError("esac_oracle_backward failed")
    out["loss"] = float(loss)
    out["entropy"] = float(out["entropy"][0])
    return out


def p3p(obj4, img4, fx, fy, cx, cy):
    obj = np.ascontiguousarray(obj4, np.float64)
    img = np.ascontiguousarray(img4, np.float64)
    r = np.zeros(3)
    t = np.zeros(3)
    ok = lib().esac_oracle_p3p(_p(obj), _p(img), fx, fy, cx, cy, _p(r), _p(t))
    return bool(ok), r, t


def p3p_all(obj3, img3, fx, fy, cx, cy):
    obj = np.ascontiguousarray(obj3, np.float64)
    img = np.ascontiguousarray(img3, np.float64)
    R = np.zeros((4, 3, 3))
    t = np.zeros((4, 3))
    n = lib().esac_oracle_p3p_all(_p(obj), _p(img), fx, fy, cx, cy, _p(R), _p(t))
    return R[:n], t[:n]


def solve_deg4(a, b, c, d, e):
    r = np.zeros(4)
    n = lib().esac_oracle_solve_deg4(a, b, c, d, e, _p(r))
    return r[:n]


def rodrigues_vec2mat(r, jac=False):
    r = np.ascontiguousarray(r, np.float64)
    R = np.zeros((3, 3))
    J = np.zeros((3, 9))
    lib().esac_oracle_rodrigues_vec2mat(_p(r), _p(R), _p(J) if jac else None)
    return (R, J) if jac else R


def rodrigues_mat2vec(R):
    R = np.ascontiguousarray(R, np.float64)
    r = np.zeros(3)
    lib().esac_oracle_rodrigues_mat2vec(_p(R), _p(r))
    return r


def rodrigues_mat2vec_svd(R):
    """cv::Rodrigues on a matrix as OpenCV runs it (nearest rotation first)."""
    R = np.ascontiguousarray(R, np.float64)
    r = np.zeros(3)
    lib().esac_oracle_rodrigues_mat2vec_svd(_p(R), _p(r))
    return r


def pose_loss(pose, gt_trans, w_rot, w_trans, cut):
    pose = np.ascontiguousarray(pose, np.float64)
    gt = np.ascontiguousarray(gt_trans, np.float64)
    return lib().esac_oracle_pose_loss(_p(pose), _p(gt), w_rot, w_trans, cut)


def pose_dloss(est, gt_pose, w_rot, w_trans, cut):
    est = np.ascontiguousarray(est, np.float64)
    gt = np.ascontiguousarray(gt_pose, np.float64)
    j = np.zeros(6)
    lib().esac_oracle_pose_dloss(_p(est), _p(gt), w_rot, w_trans, cut, _p(j))
    return j


def trans2pose(T):
    T = np.ascontiguousarray(T, np.float64)
    pose = np.zeros(6)
    lib().esac_oracle_trans2pose(_p(T), _p(pose))
    return pose


def dproject_dobj(pt, obj, rvec, tvec, focal, ppx, ppy, max_reproj):
    r = np.ascontiguousarray(rvec, np.float64)
    t = np.ascontiguousarray(tvec, np.float64)
    out = np.zeros(3)
    lib().esac_oracle_dproject_dobj(pt[0], pt[1], obj[0], obj[1], obj[2], _p(r), _p(t), focal, ppx, ppy, max_reproj, _p(out))
    return out


def norm_jac_row(rvec, tvec, focal, ppx, ppy, obj, pt, max_reproj):
    r = np.ascontiguousarray(rvec, np.float64)
    t = np.ascontiguousarray(tvec, np.float64)
    row = np.zeros(6)
    ok = lib().esac_oracle_norm_jac_row(_p(r), _p(t), focal, ppx, ppy, obj[0], obj[1], obj[2], pt[0], pt[1], max_reproj, _p(row))
    return ok, row


def pinv_sym6(A):
    A = np.ascontiguousarray(A, np.float64)
    out = np.zeros((6, 6))
    lib().esac_oracle_pinv_sym6(_p(A), _p(out))
    return out


def project(rvec, tvec, fx, fy, cx, cy, pts3):
    pts = np.ascontiguousarray(pts3, np.float32)
    uv = np.zeros((pts.shape[0], 2), np.float32)
    r = np.ascontiguousarray(rvec, np.float64)
    t = np.ascontiguousarray(tvec, np.float64)
    lib().esac_oracle_project(_p(r), _p(t), fx, fy, cx, cy, _p(pts), pts.shape[0], _p(uv))
    return uv


def lm_pnp(obj, img, fx, fy, cx, cy, pose0):
    obj = np.ascontiguousarray(obj, np.float32)
    img = np.ascontiguousarray(img, np.float32)
    pose = np.array(pose0, np.float64).copy()
    it = lib().esac_oracle_lm_pnp(_p(obj), _p(img), obj.shape[0], fx, fy, cx, cy, _p(pose))
    return pose, it


def pose2trans(pose):
    pose = np.ascontiguousarray(pose, np.float64)
    T = np.zeros((4, 4))
    lib().esac_oracle_pose2trans(_p(pose), _p(T))
    return T


def philox(ctr, key):
    c = np.asarray(ctr, np.uint32)
    k = np.asarray(key, np.uint32)
    o = np.zeros(4, np.uint32)
    lib().esac_oracle_philox4x32(_p(c), _p(k), _p(o))
    return o


def draw_cells(seed, call, hyp, tr, W, H):
    xy = np.zeros((4, 2), np.int32)
    lib().esac_oracle_draw_cells(seed, call, hyp, tr, W, H, _p(xy))
    return xy
