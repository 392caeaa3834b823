"""Host-side mirror of the reference's `esac` extension interface over the C ABI.

Reference: `PYBIND11_MODULE(..)  m.def("forward", &esac_forward); m.def("backward", &esac_backward)`
(code/esac/esac.cpp:513-516); call site test_esac.py:192-205.  `forward` below keeps
the positional signature, the in-place `outPose` write and the Python-int return
value; every failure the reference surfaces as a pybind11 `RuntimeError`
(c10::Error from `accessor<>()`, cv::Exception) is a `RuntimeError` here too.

The compute runs ONLY in libesac_hip.so (hand-written HIP for gfx950).  There is
no CPU fallback: without the library or without a HIP device the call raises.
torch is used for device memory, streams and (in distributed.py) RCCL -- never
for the arithmetic of this path.
"""
import ctypes as C
import os
import threading

import numpy as np
import torch

from . import build as _build

# ---------------------------------------------------------------- C ABI binding
RES_SCORE, RES_HYP, RES_EXPERT, RES_RVEC, RES_TVEC, RES_POSE = 0, 1, 2, 3, 6, 9
RES_REF_STEPS, RES_INLIERS, RES_PROB, RES_ENTROPY, RES_CONTENDERS, RES_LM_ITERS, RES_DOUBLES = 25, 26, 27, 28, 29, 30, 32
BUF_HYPS, BUF_SAMPLE_XY, BUF_TRIES, BUF_SCORES, BUF_RESULT = 0, 1, 2, 3, 4
BUF_INLIER_MAP, BUF_INLIER_COUNTS, BUF_WINNER_ERRS, BUF_EXACT_FLAGS, BUF_CYCLES = 5, 6, 7, 8, 9
BUF_BWD_PROBS, BUF_BWD_LOSSES, BUF_BWD_REF_HYPS, BUF_BWD_SCORE_GRADS, BUF_BWD_SLOTS, BUF_BWD_SLOT_INFO, BUF_BWD_DLOSS = \
    10, 11, 12, 13, 14, 15, 16
BUF_BWD_PATH1, BUF_BWD_PATH2 = 17, 18
BUF_REFINE_INFO = 19
BUF_BWD_TEAM_INFO = 20
BUF_SPEC_INFO = 21
BUF_SPEC_FLAGS = 22
REFINE_TEAM_MAX, REFINE_TEAM_EIGHT, REFINE_TEAM_AUTO = 32, 8, -1
REFINE_TEAM_DEFAULT = REFINE_TEAM_AUTO  # what a fresh context does: 8 members, or the smallest team <= 16 that lowers the cells per lane
MAX_REF_STEPS = 100
BWD_MAX_SLOTS = 1000

ABI_SYMBOLS = [
    "esac_hip_abi_version", "esac_hip_last_error", "esac_hip_device_count", "esac_hip_create", "esac_hip_destroy",
    "esac_hip_forward", "esac_hip_sample", "esac_hip_score", "esac_hip_select", "esac_hip_refine",
    "esac_hip_score_exact", "esac_hip_read", "esac_hip_write_hyps", "esac_hip_phase_ms", "esac_hip_set_timing",
    "esac_hip_score_span_ms", "esac_hip_forward_batch", "esac_hip_backward", "esac_hip_set_debug", "esac_hip_check",
    "esac_hip_pick_record", "esac_hip_time_stages", "esac_hip_shard_balanced", "esac_hip_set_wait",
    "esac_hip_set_refine_team", "esac_hip_host_turn",
    "esac_hip_comm_unique_id", "esac_hip_comm_init", "esac_hip_comm_destroy", "esac_hip_allreduce_sum", "esac_hip_comm_info",
    "esac_hip_host_turn_mean",
]
COMM_ID_BYTES = 128
ABI_VERSION = 6
FLAG_EXACT_SCORES, FLAG_SCORE_TILED, FLAG_SCORE_STREAM, FLAG_PACK_MAPS, FLAG_EXACT_SAMPLING, FLAG_SCORES_BY_INDEX = 1, 2, 4, 8, 16, 32
FLAG_AUTO_EXACT = 64
FLAG_REFINE_SOLO = 128
WAIT_SPIN, WAIT_YIELD, WAIT_BLOCK = 0, 1, 2
DEBUG_ERROR_IMAGE, DEBUG_COOP_STALL, DEBUG_TEAM_SPREAD, DEBUG_NO_SPECULATION, DEBUG_SPEC_SECOND_BEST, DEBUG_SPEC_LOSE_CHAIN = 1, 2, 4, 8, 16, 32


class Params(C.Structure):
    """struct esac_hip_params (include/esac_hip.h)."""
    _fields_ = [
        ("E", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("N", C.c_int32),
        ("shift_x", C.c_int32), ("shift_y", C.c_int32),
        ("focal", C.c_float), ("ppx", C.c_float), ("ppy", C.c_float),
        ("inlier_thresh", C.c_float), ("inlier_alpha", C.c_float), ("inlier_beta", C.c_float),
        ("max_reproj", C.c_float), ("sub_sampling", C.c_int32),
        ("seed", C.c_uint64), ("call", C.c_uint64),
        ("max_tries", C.c_int32), ("max_ref_steps", C.c_int32), ("hyp_offset", C.c_int32),
        ("rescore_margin", C.c_float),
        ("d_hyp_index", C.c_void_p),
        ("flags", C.c_int32),
        ("expert_base", C.c_int32),
    ]


_lib = None
_lib_lock = threading.Lock()


def load_library():
    """dlopen libesac_hip.so (built in-tree by esac_amd.build / __graft_entry__.build)."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        path = os.environ.get("ESAC_HIP_LIB", _build.LIB_PATH)  # override: A/B builds of the same ABI
        if not os.path.exists(path):
            raise RuntimeError(
                "esac: HIP extension %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for this path." % path)
        lib = C.CDLL(path)
        vp, i32, u64 = C.c_void_p, C.c_int, C.c_uint64
        pp = C.POINTER(Params)
        lib.esac_hip_abi_version.restype = i32
        lib.esac_hip_last_error.restype = C.c_char_p
        lib.esac_hip_device_count.restype = i32
        lib.esac_hip_create.argtypes = [C.POINTER(vp), i32]
        lib.esac_hip_destroy.argtypes = [vp]
        lib.esac_hip_forward.argtypes = [vp, vp, vp, pp, vp, vp, vp, vp]
        lib.esac_hip_forward_batch.argtypes = [vp, i32, vp, C.c_int64, vp, pp, vp, vp, vp, vp]
        for name in ("esac_hip_sample", "esac_hip_score", "esac_hip_select", "esac_hip_refine", "esac_hip_score_exact"):
            getattr(lib, name).argtypes = [vp, vp, vp, pp, vp]
        lib.esac_hip_backward.argtypes = [vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, pp, vp, vp]
        lib.esac_hip_read.argtypes = [vp, i32, vp, C.c_size_t]
        lib.esac_hip_write_hyps.argtypes = [vp, vp, i32]
        lib.esac_hip_phase_ms.argtypes = [vp, vp]
        lib.esac_hip_set_timing.argtypes = [vp, i32]
        lib.esac_hip_set_debug.argtypes = [vp, i32]
        lib.esac_hip_score_span_ms.argtypes = [vp, vp, vp]
        lib.esac_hip_check.argtypes = [vp]
        lib.esac_hip_pick_record.argtypes = [vp, vp, i32, vp, vp, vp, i32]
        lib.esac_hip_time_stages.argtypes = [vp, vp, vp, pp, vp, i32, vp]
        lib.esac_hip_shard_balanced.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]
        lib.esac_hip_set_wait.argtypes = [vp, i32]
        lib.esac_hip_set_refine_team.argtypes = [vp, i32]
        lib.esac_hip_host_turn.argtypes = [vp, vp]
        lib.esac_hip_host_turn_mean.argtypes = [vp, vp, i32]
        lib.esac_hip_comm_unique_id.argtypes = [vp, C.c_size_t]
        lib.esac_hip_comm_init.argtypes = [vp, i32, i32, vp, C.c_size_t]
        lib.esac_hip_comm_destroy.argtypes = [vp]
        lib.esac_hip_allreduce_sum.argtypes = [vp, vp, C.c_size_t, vp]
        lib.esac_hip_comm_info.argtypes = [vp, vp]
        for name in ABI_SYMBOLS:
            if name not in ("esac_hip_last_error",):
                getattr(lib, name).restype = i32
        lib.esac_hip_last_error.restype = C.c_char_p
        if lib.esac_hip_abi_version() != ABI_VERSION:
            raise RuntimeError("esac: libesac_hip.so ABI version mismatch")
        _lib = lib
        return lib


def _check(rc, lib):
    if rc != 0:
        raise RuntimeError("esac (HIP): %s [status %d]" % (lib.esac_hip_last_error().decode(), rc))


class Engine:
    """One device context (esac_hip_ctx): workspaces + stage entry points for one GPU."""

    def __init__(self, device=None):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("esac: no HIP device visible (torch.cuda.is_available() is False); "
                               "the MI355X path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        self.ctx = C.c_void_p()
        _check(self.lib.esac_hip_create(C.byref(self.ctx), self.device.index), self.lib)
        self._shape = None
        # the raw hipStream_t of torch's current stream: a private torch symbol (no Stream object on the per-call path)
        # with the public route as the fallback, resolved once
        self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        self._host_buf = (C.c_double * RES_DOUBLES)()  # host record of a blocking forward call
        self._host_addr = C.addressof(self._host_buf)
        self._host_np = np.frombuffer(self._host_buf, dtype=np.float64)
        self._comm = None  # (nranks, rank) once comm_init has run
        self._comm_key = None  # the process group (its ranks) the communicator mirrors (distributed.native_comm)

    def _call(self, fn, *args):
        """One C-ABI call with this engine's device current (the library calls hipSetDevice itself; the context
        manager is only needed -- and only paid for -- when torch's current device is another one)."""
        if torch.cuda.current_device() == self.device.index:
            _check(fn(self.ctx, *args), self.lib)
        else:
            with torch.cuda.device(self.device):
                _check(fn(self.ctx, *args), self.lib)

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.esac_hip_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    # -- helpers
    def _stream(self):
        # the raw hipStream_t of torch's current stream on this device
        if self._raw_stream is not None:
            return self._raw_stream(self.device.index)
        return torch.cuda.current_stream(self.device).cuda_stream

    def make_params(self, E, H, W, N, shift_x=0, shift_y=0, focal=525.0, ppx=320.0, ppy=240.0, inlier_thresh=10.0,
                    inlier_alpha=100.0, inlier_beta=0.5, max_reproj=100.0, sub_sampling=8, seed=1305, call=0,
                    max_tries=0, max_ref_steps=-1, hyp_offset=0, rescore_margin=0.0, exact_scores=False, score_shape="auto", pack_maps=False,
                    exact_sampling=False, scores_by_index=False, expert_base=0, refine_solo=False):
        p = Params()
        p.E, p.H, p.W, p.N = int(E), int(H), int(W), int(N)
        p.shift_x, p.shift_y = int(shift_x), int(shift_y)
        p.focal, p.ppx, p.ppy = float(focal), float(ppx), float(ppy)
        p.inlier_thresh, p.inlier_alpha, p.inlier_beta = float(inlier_thresh), float(inlier_alpha), float(inlier_beta)
        p.max_reproj, p.sub_sampling = float(max_reproj), int(sub_sampling)
        p.seed, p.call = int(seed) & (2**64 - 1), int(call) & (2**64 - 1)
        p.max_tries, p.max_ref_steps, p.hyp_offset = int(max_tries), int(max_ref_steps), int(hyp_offset)
        p.rescore_margin = float(rescore_margin)
        p.d_hyp_index = None
        # exact_scores: True / False, or "auto" = the guaranteed routes where they are free (ESAC_FLAG_AUTO_EXACT: what esac.forward asks for)
        p.flags = (FLAG_AUTO_EXACT if exact_scores == "auto" else FLAG_EXACT_SCORES if exact_scores else 0) | \
            {"auto": 0, "tiled": FLAG_SCORE_TILED, "stream": FLAG_SCORE_STREAM}[score_shape] | \
            (FLAG_PACK_MAPS if pack_maps else 0) | (FLAG_EXACT_SAMPLING if exact_sampling else 0) | (FLAG_SCORES_BY_INDEX if scores_by_index else 0) | \
            (FLAG_REFINE_SOLO if refine_solo else 0)
        p.expert_base = int(expert_base)
        self._shape = (int(N), int(H), int(W))
        return p

    def set_hyp_index(self, params, index_tensor):
        """Global hypothesis indices (device int32 [N]) for shards that are not a contiguous range."""
        assert index_tensor.is_cuda and index_tensor.dtype == torch.int32 and index_tensor.is_contiguous()
        params.d_hyp_index = index_tensor.data_ptr()
        self._keep_idx = index_tensor

    def _dev_inputs(self, scene_coords, hyp_assign):
        sc = scene_coords if scene_coords.is_cuda else scene_coords.to(self.device, non_blocking=True)
        ha = hyp_assign if hyp_assign.is_cuda else hyp_assign.to(self.device, non_blocking=True)
        # accessor<> honours strides (incl. the stride-0 expand() of test_esac.py:171-173); the kernels want dense
        return (sc if sc.is_contiguous() else sc.contiguous()), (ha if ha.is_contiguous() else ha.contiguous())

    # -- whole path
    def forward_device(self, scene_coords, hyp_assign, params, scores_out=None, result_out=None, want_host=True):
        """scene_coords [E,3,H,W] f32 / hyp_assign [N] i64 on this device. Returns host result (np.float64[32]) or None."""
        sc, ha = self._dev_inputs(scene_coords, hyp_assign)
        # (the library makes the context's GPU current itself: no torch device guard on this path; the host record goes
        # through one preallocated buffer whose address is known -- numpy's .ctypes costs a microsecond per call)
        rc = self.lib.esac_hip_forward(self.ctx, sc.data_ptr(), ha.data_ptr(), C.byref(params), self._stream(),
                                       scores_out.data_ptr() if scores_out is not None else None,
                                       result_out.data_ptr() if result_out is not None else None,
                                       self._host_addr if want_host else None)
        if rc != 0:
            _check(rc, self.lib)
        self._keep = (sc, ha)  # keep inputs alive until the (possibly asynchronous) kernels have run
        return self._host_np.copy() if want_host else None

    def forward_batch(self, scene_coords, hyp_assign, params, scores_out=None, result_out=None, want_host=True):
        """B frames per launch set. scene_coords [B,E,3,H,W] (or [E,3,H,W] shared by all frames), hyp_assign [B,N];
        `params` describes one frame, frame b uses call + b. Returns np.float64 [B,32] (or None)."""
        sc = scene_coords if scene_coords.is_cuda else scene_coords.to(self.device, non_blocking=True)
        ha = hyp_assign if hyp_assign.is_cuda else hyp_assign.to(self.device, non_blocking=True)
        sc, ha = sc.contiguous(), ha.contiguous()
        B = int(ha.shape[0])
        stride = int(sc.stride(0)) if sc.dim() == 5 else 0
        host = np.zeros((B, RES_DOUBLES), np.float64) if want_host else None
        self._call(self.lib.esac_hip_forward_batch, B, sc.data_ptr(), stride, ha.data_ptr(), C.byref(params), self._stream(),
                   scores_out.data_ptr() if scores_out is not None else None,
                   result_out.data_ptr() if result_out is not None else None,
                   host.ctypes.data if want_host else None)
        self._keep = (sc, ha)
        return host

    def backward_device(self, scene_coords, out_gradients, hyp_assign, gt_pose, w_rot, w_trans, loss_cut, params,
                        want_host=True):
        """Training path on this device. out_gradients: float32 [E,3,H,W] on this device, contiguous, accumulated into.
        gt_pose: 16 floats (4x4 camera pose). Returns np.float64[4] = expected loss, #refined hypotheses, entropy, 0."""
        sc, ha = self._dev_inputs(scene_coords, hyp_assign)
        if not (out_gradients.is_cuda and out_gradients.is_contiguous() and out_gradients.dtype == torch.float32
                and tuple(out_gradients.shape) == tuple(sc.shape)):
            raise RuntimeError("esac.backward: the gradient tensor must be a dense float32 device tensor shaped like sceneCoordinates")
        gt = np.ascontiguousarray(np.asarray(gt_pose, np.float32).reshape(16))
        host = np.zeros(4, np.float64) if want_host else None
        # (the library makes the context's GPU current itself: no torch device guard on the call path, as in forward_device)
        rc = self.lib.esac_hip_backward(
            self.ctx, sc.data_ptr(), out_gradients.data_ptr(), ha.data_ptr(), gt.ctypes.data,
            float(w_rot), float(w_trans), float(loss_cut), C.byref(params), self._stream(),
            host.ctypes.data if want_host else None)
        if rc != 0:
            _check(rc, self.lib)
        self._keep = (sc, ha, out_gradients)
        return host

    # -- single phases (stage-wise parity tests)
    def _phase(self, fn, scene_coords, hyp_assign, params):
        sc, ha = self._dev_inputs(scene_coords, hyp_assign)
        with torch.cuda.device(self.device):
            _check(fn(self.ctx, sc.data_ptr(), ha.data_ptr(), C.byref(params), self._stream()), self.lib)
        self._keep = (sc, ha)

    def sample(self, sc, ha, p):
        self._phase(self.lib.esac_hip_sample, sc, ha, p)

    def score(self, sc, ha, p):
        self._phase(self.lib.esac_hip_score, sc, ha, p)

    def select(self, sc, ha, p):
        self._phase(self.lib.esac_hip_select, sc, ha, p)

    def refine(self, sc, ha, p):
        self._phase(self.lib.esac_hip_refine, sc, ha, p)

    def score_exact(self, sc, ha, p):
        self._phase(self.lib.esac_hip_score_exact, sc, ha, p)

    def write_hyps(self, hyps):
        h = np.ascontiguousarray(hyps, np.float64)
        assert h.ndim == 2 and h.shape[1] == 6
        _check(self.lib.esac_hip_write_hyps(self.ctx, h.ctypes.data_as(C.c_void_p), h.shape[0]), self.lib)

    def read_slabs(self, which, k):
        """Gradient slabs [k,3,H,W] (float64) of the first k slots of the last backward call (BUF_BWD_PATH1 / _PATH2)."""
        _, H, W = self._shape
        out = np.zeros((int(k), 3, H, W), np.float64)
        _check(self.lib.esac_hip_read(self.ctx, which, out.ctypes.data_as(C.c_void_p), out.nbytes), self.lib)
        return out

    def read(self, which):
        N, H, W = self._shape or (0, 0, 0)  # (the context-level info buffers need no shape)
        shapes = {
            BUF_HYPS: ((N, 6), np.float64), BUF_SAMPLE_XY: ((N, 4, 2), np.int32), BUF_TRIES: ((N,), np.int32),
            BUF_SCORES: ((N,), np.float64), BUF_RESULT: ((RES_DOUBLES,), np.float64),
            BUF_INLIER_MAP: ((H, W), np.uint8), BUF_INLIER_COUNTS: ((MAX_REF_STEPS + 1,), np.int32),
            BUF_WINNER_ERRS: ((H, W), np.float32), BUF_EXACT_FLAGS: ((N,), np.uint8),
            BUF_CYCLES: ((32,), np.int64), BUF_REFINE_INFO: ((8,), np.int32), BUF_BWD_TEAM_INFO: ((4,), np.int32), BUF_SPEC_INFO: ((4,), np.int32), BUF_SPEC_FLAGS: ((N,), np.uint8),
            BUF_BWD_PROBS: ((N,), np.float64), BUF_BWD_LOSSES: ((N,), np.float64), BUF_BWD_REF_HYPS: ((N, 6), np.float64),
            BUF_BWD_SCORE_GRADS: ((N,), np.float64), BUF_BWD_SLOTS: ((N,), np.int32),
            BUF_BWD_SLOT_INFO: ((min(N, BWD_MAX_SLOTS), 4), np.int32), BUF_BWD_DLOSS: ((min(N, BWD_MAX_SLOTS), 6), np.float64),
        }
        shape, dt = shapes[which]
        out = np.zeros(shape, dt)
        _check(self.lib.esac_hip_read(self.ctx, which, out.ctypes.data_as(C.c_void_p), out.nbytes), self.lib)
        return out

    def time_stages(self, scene_coords, hyp_assign, params, reps=20):
        """Mean GPU time (ms) of sample / score / select / refine for this input (esac_hip_time_stages)."""
        sc, ha = self._dev_inputs(scene_coords, hyp_assign)
        out = np.zeros(4, np.float32)
        self._call(self.lib.esac_hip_time_stages, sc.data_ptr(), ha.data_ptr(), C.byref(params), self._stream(), int(reps),
                   out.ctypes.data)
        return dict(zip(("sample", "score", "select_rescore", "refine"), (float(v) for v in out)))

    def pick_record(self, records, world, zero=None):
        """Global winner among `world` per-rank records (device float64 [world*32], e.g. the tail of the all-reduced
        exchange buffer): picked on the device, returned as np.float64[32].  zero: optional device float64 tensor the same
        launch clears (the exchange buffer of the NEXT call)."""
        assert records.is_cuda and records.dtype == torch.float64 and records.is_contiguous() and records.numel() >= 32 * world
        assert zero is None or (zero.is_cuda and zero.dtype == torch.float64 and zero.is_contiguous())
        rc = self.lib.esac_hip_pick_record(self.ctx, records.data_ptr(), int(world), self._stream(), self._host_addr,
                                           zero.data_ptr() if zero is not None else None, int(zero.numel()) if zero is not None else 0)
        if rc != 0:
            _check(rc, self.lib)
        return self._host_np.copy()

    def shard_balanced(self, hyp_assign, world, rank, E, expert_base=0, index_out=None, assign_out=None, info_out=None):
        """This rank's share of the load-balanced split of `hyp_assign` (device int64 [N]), built on the device in one
        asynchronous launch (esac_hip_shard_balanced): returns (global indices int32 [n_local], local assignment int64
        [n_local], info int32 [4] = first expert, last expert, n_local, out-of-range flag) -- all device tensors."""
        assert hyp_assign.is_cuda and hyp_assign.dtype == torch.int64 and hyp_assign.is_contiguous()
        N = int(hyp_assign.shape[0])
        n_local = N // world + (1 if rank < N % world else 0)
        dev = self.device
        if index_out is None:
            index_out = torch.empty(max(n_local, 1), dtype=torch.int32, device=dev)
        if assign_out is None:
            assign_out = torch.empty(max(n_local, 1), dtype=torch.int64, device=dev)
        if info_out is None:
            info_out = torch.empty(4, dtype=torch.int32, device=dev)
        self._call(self.lib.esac_hip_shard_balanced, hyp_assign.data_ptr(), N, int(E), int(world), int(rank), int(expert_base),
                   self._stream(), index_out.data_ptr(), assign_out.data_ptr(), info_out.data_ptr())
        self._keep_shard = (hyp_assign, index_out, assign_out, info_out)
        return index_out[:n_local], assign_out[:n_local], info_out

    def set_wait(self, mode):
        """How blocking calls wait for their record: WAIT_SPIN (default), WAIT_YIELD, WAIT_BLOCK (esac_hip_set_wait)."""
        _check(self.lib.esac_hip_set_wait(self.ctx, int(mode)), self.lib)

    def check(self):
        """Waits for the device; raises if the most recent (asynchronous) call met an out-of-range hypAssignment."""
        _check(self.lib.esac_hip_check(self.ctx), self.lib)

    def set_debug(self, keep_error_image=False, coop_stall=False, team_spread=False, no_speculation=False, spec_second_best=False,
                  spec_lose_chain=False):
        _check(self.lib.esac_hip_set_debug(self.ctx, (DEBUG_ERROR_IMAGE if keep_error_image else 0) | (DEBUG_COOP_STALL if coop_stall else 0) |
                                           (DEBUG_TEAM_SPREAD if team_spread else 0) | (DEBUG_NO_SPECULATION if no_speculation else 0) |
                                           (DEBUG_SPEC_SECOND_BEST if spec_second_best else 0) | (DEBUG_SPEC_LOSE_CHAIN if spec_lose_chain else 0)), self.lib)

    def spec_info(self):
        """The speculative forward route (several experts: the straggler chain beside the refinement; ESAC_BUF_SPEC_INFO)."""
        v = self.read(BUF_SPEC_INFO)
        return {"calls": int(v[0]), "failures": int(v[1]), "last_speculative": bool(v[2]), "last_failed": bool(v[3])}

    def set_refine_team(self, members=REFINE_TEAM_DEFAULT):
        """Workgroups that share the winner's refinement on a small single-frame grid (0 / 1: one workgroup; a number: exactly
        that many; REFINE_TEAM_AUTO = the default policy: 8, or the smallest team <= 16 that lowers the cells a lane holds --
        10 on the 60x80 grid; esac_hip_set_refine_team)."""
        _check(self.lib.esac_hip_set_refine_team(self.ctx, int(members)), self.lib)

    def bwd_team_info(self):
        """Training path: were the slots of the last backward call refined by teams (ESAC_BUF_BWD_TEAM_INFO)."""
        v = self.read(BUF_BWD_TEAM_INFO)
        return {"teams": bool(v[0]), "team_calls": int(v[1]), "team_fallbacks": int(v[2]), "slots": int(v[3])}

    def refine_info(self):
        """How the most recent winner refinement ran (ESAC_BUF_REFINE_INFO).  `same_xcd`: the census of the team's first
        exchange found every member on one XCD -- the exchanges after it then stayed in that XCD's L2 (plain granule stores,
        refine_common.hpp:gran_store); otherwise they were written through (valid at any placement, 0.1-0.3 us slower each)."""
        v = self.read(BUF_REFINE_INFO)
        return {"mode": ("one_workgroup", "cooperating", "team")[int(v[0])] if 0 <= int(v[0]) <= 2 else int(v[0]),
                "workgroups": int(v[1]),
                "xcd_census": [(int(v[2]) >> (8 * x)) & 255 for x in range(4)] + [(int(v[7]) >> (8 * x)) & 255 for x in range(4)],
                "same_xcd": bool(v[3]),
                "exchanges": int(v[4]), "timed_out": bool(v[5]), "team_fallbacks": int(v[6]) & 0x3fffffff,
                "team_latched_off": bool(int(v[6]) & 0x40000000)}

    # -- the multi-GPU score exchange straight on RCCL (esac_hip_comm_*; distributed.py bootstraps the id)
    def comm_unique_id(self):
        buf = (C.c_ubyte * COMM_ID_BYTES)()
        _check(self.lib.esac_hip_comm_unique_id(buf, COMM_ID_BYTES), self.lib)
        return bytes(buf)

    def comm_init(self, nranks, rank, unique_id):
        assert len(unique_id) == COMM_ID_BYTES
        buf = (C.c_ubyte * COMM_ID_BYTES).from_buffer_copy(unique_id)
        _check(self.lib.esac_hip_comm_init(self.ctx, int(nranks), int(rank), buf, COMM_ID_BYTES), self.lib)
        self._comm = (int(nranks), int(rank))

    def comm_destroy(self):
        _check(self.lib.esac_hip_comm_destroy(self.ctx), self.lib)
        self._comm = None
        self._comm_key = None

    def comm_info(self):
        """What the communicator itself reports (esac_hip_comm_info): ranks it spans, this rank, the GPU RCCL bound it to, the
        context's GPU."""
        out = (C.c_int32 * 4)()
        _check(self.lib.esac_hip_comm_info(self.ctx, out), self.lib)
        return {"nranks": int(out[0]), "rank": int(out[1]), "rccl_device": int(out[2]), "device": int(out[3])}

    def allreduce_sum(self, buf):
        """In-place all-reduce(SUM) of a device float64 tensor over this engine's RCCL communicator, on the current stream."""
        rc = self.lib.esac_hip_allreduce_sum(self.ctx, buf.data_ptr(), int(buf.numel()), self._stream())
        if rc != 0:
            _check(rc, self.lib)

    def host_turn(self):
        """Host-side stamps of the most recent blocking forward (esac_hip_host_turn), in microseconds after entry."""
        out = np.zeros(8, np.float64)
        _check(self.lib.esac_hip_host_turn(self.ctx, out.ctypes.data), self.lib)
        return {"args_ready": out[0] * 1e-3, "sample_launched": out[1] * 1e-3, "score_launched": out[2] * 1e-3, "refine_launched": out[3] * 1e-3,
                "record_landed": out[4] * 1e-3, "returned": out[5] * 1e-3, "entry_ns": out[6]}

    def host_turn_mean(self, reset=True):
        """Means of the host-side stamps over the blocking forward calls since the last reset (esac_hip_host_turn_mean), in us:
        where the host's share of a step goes."""
        out = np.zeros(8, np.float64)
        _check(self.lib.esac_hip_host_turn_mean(self.ctx, out.ctypes.data, 1 if reset else 0), self.lib)
        u = out * 1e-3
        return {"calls": int(out[7]), "args_ready": u[0], "first_launch_call": u[1] - u[0], "further_launch_calls": u[3] - u[1],
                "wait_for_record": u[4] - u[3], "record_to_return": u[5] - u[4], "call_total": u[5], "between_calls": u[6]}

    def set_timing(self, on, period=1):
        """Per-phase events on every `period`-th forward call (the next call is the first sampled one)."""
        _check(self.lib.esac_hip_set_timing(self.ctx, (max(1, int(period)) if on else 0)), self.lib)

    def phase_ms(self):
        out = np.zeros(6, np.float32)
        _check(self.lib.esac_hip_phase_ms(self.ctx, out.ctypes.data_as(C.c_void_p)), self.lib)
        return out

    def score_span_ms(self):
        """(mean device-side duration of the score kernel in ms, number of launches averaged)."""
        ms, n = C.c_float(0), C.c_int(0)
        _check(self.lib.esac_hip_score_span_ms(self.ctx, C.byref(ms), C.byref(n)), self.lib)
        return float(ms.value), int(n.value)


# ---------------------------------------------------------------- module-level state
# The reference keeps a static RNG whose state advances from call to call
# (thread_rand.cpp:4-5); here that state is (seed, call counter).
_state = {"seed": 1305, "call": 0, "engines": {}, "last": None, "max_tries": 0, "max_ref_steps": -1, "fwd_cache": {},
          "exact_scores": None, "exact_sampling": False}


def set_seed(seed, call=0):
    """ThreadRand::forceInit equivalent (thread_rand.cpp:7-11; not exported by the reference)."""
    _state["seed"], _state["call"] = int(seed), int(call)


def get_rng_state():
    return _state["seed"], _state["call"]


def set_limits(max_tries=0, max_ref_steps=-1):
    """Override MAX_SAMPLING_TRIES / MAX_REF_STEPS (esac.cpp:44-45); 0 / -1 restore the reference values."""
    _state["max_tries"], _state["max_ref_steps"] = int(max_tries), int(max_ref_steps)


def set_exact_scores(on):
    """True: every hypothesis is scored in the reference's arithmetic (ESAC_FLAG_EXACT_SCORES), so the score vector of
    last_result() and the record's probability / entropy are the reference's own values; the pose is the same either way.
    False: the fp32 ranking stream + exact re-score of the contenders everywhere.  None (the default): exact where it is
    free -- one expert, N * H * W <= 2^21, i.e. the reference's own 64- and 256-hypothesis configurations
    (ESAC_FLAG_AUTO_EXACT) -- and the ranking stream elsewhere."""
    _state["exact_scores"] = None if on is None else bool(on)


def set_exact_sampling(on):
    """True: no screen in the sampling loop -- every try of every hypothesis is solved and decided by the fp64 route
    (ESAC_FLAG_EXACT_SAMPLING), the reference's loop try by try (esac_util.h:152-223).  The accepted try is the same either
    way; this is the guaranteed route (several times slower on wrong-expert hypotheses)."""
    _state["exact_sampling"] = bool(on)


def engine(device=None):
    idx = torch.cuda.current_device() if device is None else int(device)
    eng = _state["engines"].get(idx)
    if eng is None:
        eng = _state["engines"][idx] = Engine(idx)
    return eng


def last_result():
    """Details of the most recent forward(): scores tensor (device, float64 [N]; a buffer the next call with the same
    signature overwrites -- clone it to keep it) and the result record."""
    return _state["last"]


def _validate(sceneCoordinates, hypAssignment, outPose):
    # what accessor<float,4>() / accessor<long,1>() / accessor<float,2>() enforce (esac.cpp:80-84,184)
    for name, t, dt, nd in (("sceneCoordinates", sceneCoordinates, torch.float32, 4),
                            ("hypAssignment", hypAssignment, torch.int64, 1), ("outPose", outPose, torch.float32, 2)):
        if not isinstance(t, torch.Tensor):
            raise RuntimeError("esac.forward: %s must be a torch.Tensor" % name)
        if t.dtype != dt:
            raise RuntimeError("esac.forward: expected scalar type %s for %s but found %s" % (dt, name, t.dtype))
        if t.dim() != nd:
            raise RuntimeError("esac.forward: expected %d dims for %s but tensor has %d" % (nd, name, t.dim()))
    if sceneCoordinates.size(1) != 3:
        raise RuntimeError("esac.forward: sceneCoordinates must be [E,3,H,W]")
    if tuple(outPose.shape) != (4, 4):
        raise RuntimeError("esac.forward: outPose must be [4,4]")
    if hypAssignment.numel() == 0:
        raise RuntimeError("esac.forward: hypAssignment is empty")


def forward(sceneCoordinates, hypAssignment, outPose, shiftX, shiftY, focalLength, ppointX, ppointY,
            inlierThreshold, inlierAlpha, inlierBeta, maxReproj, subSampling):
    """Drop-in for `esac.forward` (esac.cpp:64-77): estimates the pose, writes the 4x4 camera
    transform into `outPose` in place and returns the winning expert as a Python int.

    Tensors may live on the CPU (as the reference requires) or already on the GPU
    (then the `.cpu()` at test_esac.py:187 can be dropped)."""
    _validate(sceneCoordinates, hypAssignment, outPose)
    dev = sceneCoordinates.device.index if sceneCoordinates.is_cuda else None
    eng = engine(dev)
    E, _, H, W = sceneCoordinates.shape
    N = hypAssignment.shape[0]
    if not hypAssignment.is_cuda:
        lo, hi = torch.aminmax(hypAssignment)
        if int(lo) < 0 or int(hi) >= E:
            raise RuntimeError("esac.forward: hypAssignment values must lie in [0,%d), found [%d,%d]" % (E, int(lo), int(hi)))
    # parameter block and score buffer are kept per SHAPE; the scalar fields are rewritten per call (a per-frame focal
    # length -- Aachen, Dubrovnik -- must not evict anything)
    key = (eng.device.index, E, H, W, N)
    cached = _state["fwd_cache"].get(key)
    if cached is None:
        if len(_state["fwd_cache"]) > 16:
            _state["fwd_cache"].clear()
        cached = [eng.make_params(E, H, W, N), torch.empty(N, dtype=torch.float64, device=eng.device), None]
        _state["fwd_cache"][key] = cached
    p, scores = cached[0], cached[1]
    if not sceneCoordinates.is_cuda or not hypAssignment.is_cuda:
        # the reference's convention (test_esac.py:187 `.cpu()`): CPU tensors in.  They go through PINNED staging buffers kept per
        # shape -- one host copy (which also resolves strides and the stride-0 expand() of --expertselection) and an asynchronous
        # H2D on the launch stream, instead of a pageable-memory transfer and a device allocation per call.  The call is
        # blocking, so the buffers are free again when it returns.
        if cached[2] is None:
            cached[2] = (torch.empty((E, 3, H, W), dtype=torch.float32, pin_memory=True), torch.empty((E, 3, H, W), dtype=torch.float32, device=eng.device),
                         torch.empty(N, dtype=torch.int64, pin_memory=True), torch.empty(N, dtype=torch.int64, device=eng.device), None)
        pin_sc, dev_sc, pin_ha, dev_ha = cached[2][:4]
        with torch.cuda.device(eng.device):
            if not sceneCoordinates.is_cuda:
                pin_sc.copy_(sceneCoordinates)
                dev_sc.copy_(pin_sc, non_blocking=True)
                sceneCoordinates = dev_sc
            if not hypAssignment.is_cuda:
                if E == 1:
                    # one expert: the kernels never read the assignment vector (device_common.hpp:expert_of), and the host check
                    # above has seen that it holds nothing but zeros -- no transfer
                    if cached[2][4] is None:
                        cached[2] = cached[2][:4] + (torch.zeros(N, dtype=torch.int64, device=eng.device),)
                    hypAssignment = cached[2][4]
                else:
                    pin_ha.copy_(hypAssignment)
                    dev_ha.copy_(pin_ha, non_blocking=True)
                    hypAssignment = dev_ha
    p.shift_x, p.shift_y = int(shiftX), int(shiftY)
    p.focal, p.ppx, p.ppy = float(focalLength), float(ppointX), float(ppointY)
    p.inlier_thresh, p.inlier_alpha, p.inlier_beta = float(inlierThreshold), float(inlierAlpha), float(inlierBeta)
    p.max_reproj, p.sub_sampling = float(maxReproj), int(subSampling)
    p.max_tries, p.max_ref_steps = int(_state["max_tries"]), int(_state["max_ref_steps"])
    p.flags = (FLAG_AUTO_EXACT if _state["exact_scores"] is None else FLAG_EXACT_SCORES if _state["exact_scores"] else 0) | \
        (FLAG_EXACT_SAMPLING if _state["exact_sampling"] else 0)
    p.seed, p.call = _state["seed"] & (2**64 - 1), _state["call"] & (2**64 - 1)
    eng._shape = (int(N), int(H), int(W))
    _state["call"] += 1
    res = eng.forward_device(sceneCoordinates, hypAssignment, p, scores_out=scores)
    # in place, caller-owned (esac.cpp:184-187)
    if not outPose.is_cuda and outPose.is_contiguous():
        outPose.numpy()[:] = res[RES_POSE:RES_POSE + 16].reshape(4, 4)
    else:
        outPose.copy_(torch.from_numpy(res[RES_POSE:RES_POSE + 16].astype(np.float32).reshape(4, 4)))
    _state["last"] = {"scores": scores, "result": res, "winner": int(res[RES_HYP]), "expert": int(res[RES_EXPERT])}
    return int(res[RES_EXPERT])


def forward_batch(sceneCoordinates, hypAssignment, outPoses, shiftX, shiftY, focalLength, ppointX, ppointY,
                  inlierThreshold, inlierAlpha, inlierBeta, maxReproj, subSampling):
    """Batched companion of `forward` (new API): sceneCoordinates [B,E,3,H,W] (or [E,3,H,W] shared),
    hypAssignment [B,N] int64, outPoses [B,4,4] float32 written in place; returns the list of winning experts.
    Frame b is what the b-th of B consecutive `forward` calls would compute (discrete outputs identical; poses to the rounding of
    the LM sums, bit for bit when the single calls refine with teams of 8 like a batch does: include/esac_hip.h)."""
    if hypAssignment.dim() != 2 or hypAssignment.dtype != torch.int64:
        raise RuntimeError("esac.forward_batch: hypAssignment must be int64 [B,N]")
    if sceneCoordinates.dtype != torch.float32 or sceneCoordinates.dim() not in (4, 5) or sceneCoordinates.size(-3) != 3:
        raise RuntimeError("esac.forward_batch: sceneCoordinates must be float32 [B,E,3,H,W] or [E,3,H,W]")
    B, N = hypAssignment.shape
    if outPoses.dtype != torch.float32 or tuple(outPoses.shape) != (B, 4, 4):
        raise RuntimeError("esac.forward_batch: outPoses must be float32 [B,4,4]")
    if sceneCoordinates.dim() == 5 and sceneCoordinates.size(0) != B:
        raise RuntimeError("esac.forward_batch: batch sizes of sceneCoordinates and hypAssignment differ")
    eng = engine(sceneCoordinates.device.index if sceneCoordinates.is_cuda else None)
    E, H, W = sceneCoordinates.shape[-4], sceneCoordinates.shape[-2], sceneCoordinates.shape[-1]
    p = eng.make_params(E, H, W, N, shiftX, shiftY, focalLength, ppointX, ppointY, inlierThreshold, inlierAlpha,
                        inlierBeta, maxReproj, subSampling, seed=_state["seed"], call=_state["call"],
                        max_tries=_state["max_tries"], max_ref_steps=_state["max_ref_steps"])
    _state["call"] += B
    scores = torch.empty(B, N, dtype=torch.float64, device=eng.device)
    res = eng.forward_batch(sceneCoordinates, hypAssignment, p, scores_out=scores)
    outPoses.copy_(torch.from_numpy(res[:, RES_POSE:RES_POSE + 16].astype(np.float32).reshape(B, 4, 4)))
    _state["last"] = {"scores": scores, "result": res}
    return [int(v) for v in res[:, RES_EXPERT]]


def backward(sceneCoordinates, outGradients, hypAssignment, gtPose, wLossRot, wLossTrans, lossCut, shiftX, shiftY,
             focalLength, ppointX, ppointY, inlierThreshold, inlierAlpha, inlierBeta, maxReproj, subSampling):
    """Drop-in for `esac.backward` (esac.cpp:213-230): expected pose loss over the hypothesis distribution; its
    gradient wrt the scene coordinates is ADDED to `outGradients` in place (esac.cpp:491-508, the caller passes
    zeros: train_esac.py:148). Returns the expected loss as a Python float.

    Tensors may live on the CPU (as train_esac.py:152-155 passes them) or on the GPU; with device tensors nothing
    but the ground-truth pose and the loss value crosses PCIe."""
    if sceneCoordinates.dtype != torch.float32 or sceneCoordinates.dim() != 4 or sceneCoordinates.size(1) != 3:
        raise RuntimeError("esac.backward: sceneCoordinates must be float32 [E,3,H,W]")
    if outGradients.dtype != torch.float32 or tuple(outGradients.shape) != tuple(sceneCoordinates.shape):
        raise RuntimeError("esac.backward: outGradients must be float32 and shaped like sceneCoordinates")
    if hypAssignment.dtype != torch.int64 or hypAssignment.dim() != 1 or hypAssignment.numel() == 0:
        raise RuntimeError("esac.backward: hypAssignment must be a non-empty int64 [N]")
    if gtPose.dtype != torch.float32 or tuple(gtPose.shape) != (4, 4):
        raise RuntimeError("esac.backward: gtPose must be float32 [4,4]")
    dev = sceneCoordinates.device.index if sceneCoordinates.is_cuda else None
    eng = engine(dev)
    E, _, H, W = sceneCoordinates.shape
    N = hypAssignment.shape[0]
    if not hypAssignment.is_cuda:
        lo, hi = int(hypAssignment.min()), int(hypAssignment.max())
        if lo < 0 or hi >= E:
            raise RuntimeError("esac.backward: hypAssignment values must lie in [0,%d), found [%d,%d]" % (E, lo, hi))
    p = eng.make_params(E, H, W, N, shiftX, shiftY, focalLength, ppointX, ppointY, inlierThreshold, inlierAlpha,
                        inlierBeta, maxReproj, subSampling, seed=_state["seed"], call=_state["call"],
                        max_tries=_state["max_tries"], max_ref_steps=_state["max_ref_steps"])
    _state["call"] += 1
    in_place = outGradients.is_cuda and outGradients.is_contiguous() and outGradients.device == eng.device
    grads = outGradients if in_place else outGradients.to(eng.device).contiguous()
    out = eng.backward_device(sceneCoordinates, grads, hypAssignment, gtPose.detach().cpu().numpy(), wLossRot, wLossTrans,
                              lossCut, p)
    if not in_place:
        outGradients.copy_(grads)  # the accumulated tensor back into the caller's (CPU or strided) storage
    _state["last"] = {"backward": out}
    return float(out[0])
