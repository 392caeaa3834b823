"""Hypothesis sharding across GPUs + the single score all-reduce.

The reference has no multi-device path (no NCCL/MPI anywhere); this is new design.
Hypotheses are independent through sampling and scoring (esac_util.h:152-225 and
esac.cpp:131-147 only touch slot [h]) and the output depends on non-winners only
through the argmax, so each rank
  1. scores its own shard and refines its LOCAL best (refining non-winners cannot
     change the result, esac.cpp:167-177 refines the winner only),
  2. contributes one zero-padded buffer  [ N_total scores | world x 32-double records ]
     to ONE all-reduce(SUM) (RCCL over xGMI on GPUs, gloo in the CPU tests),
  3. picks the global winner = max exact score, lowest GLOBAL hypothesis index on
     ties (esac_util.h:519 "first max").
RNG streams and tie-breaks use global hypothesis indices, so the result does not
depend on the number of ranks.  Which hypotheses a rank takes: `policy="balanced"` (default for
several experts) orders them by (expert, index) and cuts that order into `world` equal pieces --
N/world hypotheses per rank whatever the gating distribution, a contiguous range of experts per rank,
only the experts at the cuts shared with a neighbour; the shard is built ON THE DEVICE from the
assignment vector (esac_hip_shard_balanced: one launch, no host round trip) and the kernels write the
scores through the global index straight into the exchange buffer.  `policy="range"` cuts the index
range (every rank needs every map); `policy="expert"` (expert e on rank e % world) is kept for
comparison -- it does not balance (cfg4 on 4 ranks: 3992 of 4096 hypotheses on one rank).  The payload is (N_total + 32*world) * 8 bytes
(<= 133 KB at N = 16384, 8 ranks): latency-bound, xGMI bandwidth is irrelevant.
"""
import time

import numpy as np
import torch
import torch.distributed as dist

RES_SCORE, RES_HYP, RES_DOUBLES = 0, 1, 32
RES_VALID = RES_DOUBLES - 1  # slot 31 of a device record: 1 = a record, 0 = none (empty shard), 3 = the rank's refinement team timed out


class TeamTimeout(RuntimeError):
    """The refinement team of at least one rank did not synchronise (ESAC_RES_VALID = 3 in its record, status -12 of the pick):
    no winner was declared.  Every rank reads the same all-reduced records, so every rank raises this for the same frame."""


def shard_range(n_total, rank, world):
    """Contiguous index-range shard [lo, hi) of rank `rank`."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_expert(hyp_assign, rank, world):
    """Global indices of the hypotheses whose expert this rank owns (expert e lives on rank e % world):
    each rank then only needs its own experts' maps (and only runs those expert CNNs)."""
    ha = np.asarray(hyp_assign)
    return np.nonzero(ha % world == rank)[0].astype(np.int32)


def shard_balanced_host(hyp_assign, rank, world, E=None):
    """CPU mirror of esac_hip_shard_balanced (the plan is a pure function of the assignment vector): the global indices
    of rank `rank`'s share, in (expert, index) order, and its (first, last) expert.  Values outside [0, E) count as
    expert 0, like device_common.hpp:expert_of."""
    e = np.asarray(hyp_assign).astype(np.int64).copy()
    if E is not None:
        e[(e < 0) | (e >= E)] = 0
    order = np.argsort(e, kind="stable")
    lo, hi = shard_range(len(e), rank, world)
    idx = order[lo:hi].astype(np.int32)
    rng = (int(e[idx[0]]), int(e[idx[-1]])) if len(idx) else (0, -1)
    return idx, rng


def plan_balanced(counts, world):
    """Expert range (first, last) of every rank under policy "balanced", from the hypothesis histogram the caller takes
    anyway (test_esac.py:178 `torch.histc`): what a rank must hold maps for / run expert CNNs for.  (0, -1): no share."""
    counts = np.asarray(counts, np.int64)
    ends = np.cumsum(counts)
    starts = ends - counts
    n = int(ends[-1]) if len(ends) else 0
    out = []
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        if hi <= lo:
            out.append((0, -1))
            continue
        first = int(np.searchsorted(ends, lo, side="right"))      # expert containing sorted position lo
        last = int(np.searchsorted(ends, hi - 1, side="right"))   # ... position hi - 1
        out.append((first, last))
    return out


def pack_local(scores_local, record_local, n_total, global_index, rank, world):
    """Zero-padded contribution of one rank.  scores_local [n_local] f64, record_local [32] f64 (same device)."""
    buf = torch.zeros(n_total + world * RES_DOUBLES, dtype=torch.float64, device=scores_local.device)
    if scores_local.numel():
        buf[global_index.to(scores_local.device, torch.long)] = scores_local
        rec = record_local.to(torch.float64)
        # +1 marker in the last slot: a rank with an empty shard contributes an all-zero record
        rec = rec.clone()
        rec[RES_DOUBLES - 1] = 1.0
        buf[n_total + rank * RES_DOUBLES: n_total + (rank + 1) * RES_DOUBLES] = rec
    return buf


def pick_global(buf, n_total, world, engine=None, zero=None):
    """(scores_global [n_total], winning record [32]) from the all-reduced buffer.  With a device buffer and an engine
    the pick runs on the device (esac_hip_pick_record: one launch, the record lands in pinned host memory; the same launch
    clears `zero`, the exchange buffer of the next call); a CPU buffer (gloo tests) is scanned on the host."""
    scores = buf[:n_total]
    if engine is not None and buf.is_cuda:
        try:
            return scores, engine.pick_record(buf[n_total:], world, zero=zero)
        except RuntimeError as exc:
            if "no rank produced" in str(exc):
                raise RuntimeError("esac: no rank produced a hypothesis")
            if "[status -12]" in str(exc):
                raise TeamTimeout(str(exc))
            raise
    recs = buf[n_total:].view(world, RES_DOUBLES).cpu().numpy()
    if (recs[:, RES_VALID] == 3.0).any():
        raise TeamTimeout("esac: the refinement team of rank(s) %s timed out" % np.nonzero(recs[:, RES_VALID] == 3.0)[0].tolist())
    best = None
    for r in range(world):
        rec = recs[r]
        if rec[RES_VALID] != 1.0:
            continue  # empty shard
        if best is None or rec[RES_SCORE] > best[RES_SCORE] or (rec[RES_SCORE] == best[RES_SCORE] and rec[RES_HYP] < best[RES_HYP]):
            best = rec
    if best is None:
        raise RuntimeError("esac: no rank produced a hypothesis")
    return scores, best.copy()


_buffers = {}


class _Exchange:
    """The all-reduce payload of one (device, N, world): TWO persistent buffers that alternate call by call.  A rank's kernels
    write only its own slots, the all-reduce then fills every slot -- so a buffer must be all-zero outside the rank's slots
    when a call starts.  Instead of a memset in front of every call, the winner-pick launch that ends call i clears the
    buffer call i + 1 will use (esac_hip_pick_record's d_zero): a call is the forward launches, the collective and the pick,
    nothing else.  (The score view returned by call i therefore also survives call i + 1.)"""

    def __init__(self, dev, n_total, world):
        self.bufs = [torch.zeros(n_total + world * RES_DOUBLES, dtype=torch.float64, device=dev) for _ in range(2)]
        self.clean = [True, True]  # all-zero and not handed out since
        self.turn = 0

    def take(self, need_zero=True):
        """(this call's buffer, the next call's buffer).  A buffer is clean when the pick launch of the previous call has cleared
        it (`cleared`); a call that ended early -- an exception between here and its pick -- leaves the other one holding the
        all-reduced data of the call before, and the next taker clears it itself.  need_zero=False: the caller overwrites every
        slot (one rank: the whole buffer is its own)."""
        i = self.turn
        cur, nxt = self.bufs[i], self.bufs[i ^ 1]
        if need_zero and not self.clean[i]:
            cur.zero_()
        self.clean[i] = False
        self.turn ^= 1
        self._next = i ^ 1
        return cur, nxt

    def cleared(self):
        """The pick launch of the call that took last has run with zero=<its next buffer>."""
        self.clean[self._next] = True


def _exchange(dev, n_total, world):
    key = ("pair", str(dev), n_total, world)
    ex = _buffers.get(key)
    if ex is None:
        ex = _buffers[key] = _Exchange(dev, n_total, world)
    return ex


def _exchange_buffer(dev, n_total, world):
    """One persistent all-reduce payload per (device, N, world) for the index-list policies (zeroed by the caller per call)."""
    key = (str(dev), n_total, world)
    buf = _buffers.get(key)
    if buf is None:
        buf = torch.zeros(n_total + world * RES_DOUBLES, dtype=torch.float64, device=dev)
        _buffers[key] = buf
    return buf


_params_cache = {}


def _shard_params(engine, E, H, W, n, lo, params_kw):
    """Parameter block of a shard; per frame only the RNG key moves, so the block is built once and re-keyed."""
    fixed = tuple(sorted((k, v) for k, v in params_kw.items() if k not in ("seed", "call")))
    key = (id(engine), E, H, W, n, lo, fixed)
    p = _params_cache.get(key)
    if p is None:
        if len(_params_cache) > 64:
            _params_cache.clear()
        p = engine.make_params(E, H, W, n, hyp_offset=lo, **params_kw)
        _params_cache[key] = p
    else:
        p.seed = int(params_kw.get("seed", 1305)) & (2**64 - 1)
        p.call = int(params_kw.get("call", 0)) & (2**64 - 1)
        engine._shape = (int(n), int(H), int(W))
    return p


def contribute_range(engine, scene_coords, hyp_assign_full, params_kw, rank, world, buf, zero=True, want_host=False):
    """Rank `rank`'s part of the exchange for contiguous-range sharding: run the forward path on hypotheses [lo, hi) with
    the kernels writing scores and record into this rank's slots of `buf` (zeroed first unless the caller knows it is:
    forward_sharded's alternating pair).  want_host: also hand the rank's own record back (a blocking call)."""
    n_total = int(hyp_assign_full.shape[0])
    E, _, H, W = scene_coords.shape
    lo, hi = shard_range(n_total, rank, world)
    if zero:
        buf.zero_()
    rec = None
    if hi > lo:
        dev = engine.device
        ha_dev = hyp_assign_full if hyp_assign_full.is_cuda else hyp_assign_full.to(dev)
        if ha_dev.stride(0) != 1:
            ha_dev = ha_dev.contiguous()  # stride-0 expand() of --expertselection
        p = _shard_params(engine, E, H, W, hi - lo, lo, params_kw)
        rec0 = n_total + rank * RES_DOUBLES
        # the refinement kernel writes the record itself, incl. the "this rank contributed" marker in its last slot
        # (ESAC_RES_VALID, the convention pack_local follows for the index-list path)
        rec = engine.forward_device(scene_coords, ha_dev[lo:hi], p, scores_out=buf[lo:hi], result_out=buf[rec0:rec0 + RES_DOUBLES],
                                    want_host=want_host)
    return (buf, rec) if want_host else buf


_native = {"off": False}
_WORLD = "the default process group"


def native_comm(engine, group=None):
    """This engine's rank in an RCCL communicator of the LIBRARY's own (esac_hip_comm_init), mirroring `group`: rank 0 makes the
    unique id, one broadcast over the torch.distributed group hands it round (control plane, once), every rank joins.  From then on
    the per-frame collective is one C call that enqueues ncclAllReduce on the launch stream -- torch.distributed's own enqueue
    of a collective costs the host 20-27 us per call (bench.py: sharded_world1).  Only for backend "nccl" (one rank per GPU);
    None when the group is gloo (CPU tests, several ranks sharing one GPU), when ESAC_NATIVE_RCCL=0, or when joining failed (the
    torch.distributed all-reduce -- RCCL as well -- then carries the exchange).
    Every rank runs the SAME sequence of control-plane collectives whatever fails where: (1) every rank probes that the library
    can bind RCCL (rank 0's probe IS the id), (2) broadcast of (id or None), (3) MIN over "I can join" -- only then
    ncclCommInitRank, which blocks until all ranks are in it -- (4) MIN over "I joined"."""
    import os
    if _native["off"] or os.environ.get("ESAC_NATIVE_RCCL", "1") == "0" or not dist.is_initialized() or dist.get_backend(group) != "nccl":
        return None
    if engine._comm is not None and engine._comm_key is not None and engine._comm_key[0] is (group if group is not None else _WORLD):
        return engine  # (per frame: one identity test -- the group object the communicator was made for)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    key = (group if group is not None else _WORLD, tuple(dist.get_process_group_ranks(group if group is not None else dist.group.WORLD)))
    why = None
    my_id = None
    try:
        my_id = engine.comm_unique_id()  # also the probe that RCCL can be bound in this process
    except Exception as exc:
        why = str(exc)
    box = [my_id if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if box[0] is None and why is None:
        why = "rank 0 could not make an RCCL unique id"

    def all_agree(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=engine.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return int(flag.item()) == 1

    joined = False
    if all_agree(why is None):
        try:
            engine.comm_init(world, rank, box[0])
            engine._comm_key = key
            joined = True
        except Exception as exc:
            why = str(exc)
        if all_agree(joined):
            return engine
    import warnings
    warnings.warn("esac: RCCL communicator of the library could not be set up on every rank (%s); using torch.distributed.all_reduce"
                  % (why or "another rank failed"))
    if joined:
        engine.comm_destroy()
    _native["off"] = True
    return None


def _all_reduce_sum(buf, group, timers=None, engine=None):
    """The one collective: all-reduce(SUM) of the device buffer in place, over RCCL -- through the library's own communicator
    (native_comm) when there is one, else through torch.distributed ("nccl" IS RCCL on ROCm); a gloo group (CPU tests, or
    several ranks sharing one GPU) gets the 1-2 KB payload staged through the host.  `timers`: optional list that receives
    ("allreduce", (start, end)) CUDA event pairs around the collective and the host's time in the call (bench.py)."""
    ev = None
    if timers is not None and buf.is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    t_host = time.perf_counter() if timers is not None else 0.0
    nat = native_comm(engine, group) if engine is not None and buf.is_cuda else None
    if nat is not None:
        nat.allreduce_sum(buf)
    elif buf.is_cuda and dist.get_backend(group) == "gloo":
        host = buf.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        buf.copy_(host)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    if timers is not None:
        timers.append(("allreduce_host_ms", (time.perf_counter() - t_host) * 1e3))  # the host's time inside the call (enqueue)
    if ev is not None:
        ev[1].record()
        timers.append(("allreduce", ev))


def owned_experts(E, rank, world):
    """Experts whose maps rank `rank` holds under policy "expert": e % world == rank (local index e // world)."""
    return list(range(rank, E, world))


def forward_sharded(engine, scene_coords, hyp_assign_full, params_kw, group=None, policy="range", maps="full", timers=None):
    """forward_sharded_once, and once more with every rank refining in ONE workgroup (ESAC_FLAG_REFINE_SOLO) when a rank's
    refinement team timed out: the calls of a multi-rank exchange are asynchronous (no rank waits for its own record before the
    collective), so the blocking call's own recovery does not apply -- the failed record travels through the all-reduce
    (ESAC_RES_VALID = 3), the pick refuses to declare a winner (-12) on EVERY rank alike, and every rank comes back here.  The
    rank whose team it was counts the strike (esac_hip_check): two in a row and its context stops asking for teams."""
    try:
        return forward_sharded_once(engine, scene_coords, hyp_assign_full, params_kw, group, policy, maps, timers)
    except TeamTimeout:
        try:
            engine.check()  # (-12 on the rank whose team failed: counted towards the latch there; other ranks pass)
        except RuntimeError:
            pass
        return forward_sharded_once(engine, scene_coords, hyp_assign_full, dict(params_kw, refine_solo=True), group, policy, maps, timers)


def forward_sharded_once(engine, scene_coords, hyp_assign_full, params_kw, group=None, policy="range", maps="full", timers=None):
    """Multi-GPU esac_forward: every rank holds `scene_coords` (or only its experts' maps) and the full assignment
    vector; returns (scores_global [N] f64 device tensor, winning record np[32]).

    `params_kw` are the keyword arguments of Engine.make_params except N / hyp_offset.
    policy "range": contiguous index ranges -- the kernels write this rank's scores and record straight into its
    slots of the persistent exchange buffer (hyp_offset keys RNG and tie-breaks), so a call is one memset, the
    forward launches, the all-reduce and the winner pick.
    policy "balanced": every rank takes N / world hypotheses of a contiguous expert range, shard built on the device
    per frame (contribute_balanced); maps="owned": `scene_coords` holds the maps of params_kw["expert_range"] only.
    policy "expert": shard by expert ownership (expert e lives on rank e % world; index lists).  maps="owned":
    `scene_coords` is [E_local,3,H,W] holding ONLY this rank's experts (owned_experts(E, rank, world), in that
    order; pass `E` through params_kw["total_experts"]) -- what BASELINE configs[3]/[4] describe: a rank runs and
    stores only its own experts."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    ha_full = hyp_assign_full
    n_total = int(ha_full.shape[0])
    params_kw = dict(params_kw)
    total_experts = params_kw.pop("total_experts", None)
    E, _, H, W = scene_coords.shape
    dev = engine.device
    if policy == "range":
        # the returned score vector is a view of a persistent buffer: valid until the call after the next on this device
        ex = _exchange(dev, n_total, world)
        buf, nxt = ex.take(need_zero=world > 1)
        if world == 1:
            # one rank: its record IS the winner -- the refinement kernel hands it to the host itself (no pick launch, no second
            # wait); the collective still runs (a one-rank RCCL all-reduce on the launch stream: the dtype / stream path N ranks take)
            _, rec = contribute_range(engine, scene_coords, ha_full, params_kw, 0, 1, buf, zero=False, want_host=True)
            if dist.is_initialized():
                _all_reduce_sum(buf, group, timers, engine)
            if rec is None:
                raise RuntimeError("esac: no rank produced a hypothesis")
            return buf[:n_total], rec
        contribute_range(engine, scene_coords, ha_full, params_kw, rank, world, buf, zero=False)
        if dist.is_initialized():
            _all_reduce_sum(buf, group, timers, engine)  # the one collective of this path
        return _pick_and_clear(ex, buf, nxt, n_total, world, engine)
    if policy == "balanced":
        return _forward_balanced(engine, scene_coords, ha_full, params_kw, total_experts, rank, world, group, maps, timers)
    if policy != "expert":
        raise ValueError(policy)
    owned = maps == "owned"
    if owned and total_experts is None:
        raise ValueError('maps="owned" needs params_kw["total_experts"]')
    # keyed by the tensor AND its version counter: a caller that refills a preallocated assignment tensor in place must not
    # get the previous frame's index list
    key = ("expert", str(dev), n_total, world, rank, owned, int(ha_full.data_ptr()) if ha_full.is_cuda else id(ha_full), int(ha_full._version))
    shard = _shard_cache.get(key)
    if shard is None:  # the shard of an assignment vector: index list + local assignment, built once per vector
        while len(_shard_cache) >= 8:  # a fresh tensor per frame always misses: keep only a few entries alive
            _shard_cache.pop(next(iter(_shard_cache)))
        gidx = torch.from_numpy(shard_by_expert(ha_full.cpu().numpy(), rank, world))
        gidx_dev = gidx.to(dev).contiguous()
        ha_local = ha_full.to(dev)[gidx_dev.to(torch.long)].contiguous() if gidx.numel() else torch.empty(0, dtype=torch.int64, device=dev)
        if owned:
            ha_local = torch.div(ha_local, world, rounding_mode="floor")  # e -> its index among this rank's maps
        shard = _shard_cache[key] = (gidx_dev, ha_local, ha_full)
    gidx_dev, ha_local, _ = shard
    n_local = int(gidx_dev.numel())
    buf = _exchange_buffer(dev, n_total, world)
    buf.zero_()
    if n_local > 0:
        p = engine.make_params(E, H, W, n_local, **params_kw)
        engine.set_hyp_index(p, gidx_dev)
        scores_local = torch.empty(n_local, dtype=torch.float64, device=dev)
        rec0 = n_total + rank * RES_DOUBLES
        record = buf[rec0:rec0 + RES_DOUBLES]  # the refinement kernel writes record + ESAC_RES_VALID marker in place
        engine.forward_device(scene_coords, ha_local, p, scores_out=scores_local, result_out=record, want_host=False)
        buf[gidx_dev.to(torch.long)] = scores_local
        if owned:
            record[2] = record[2] * world + rank  # local map index -> global expert id (esac.cpp:189 returns it)
    if dist.is_initialized():  # (also in a one-rank group: the collective's dtype / stream path is then the one N ranks take)
        _all_reduce_sum(buf, group, timers, engine)  # the one collective of this path
    return pick_global(buf, n_total, world, engine)


_shard_cache = {}
_balanced_ws = {}


def _forward_balanced(engine, scene_coords, ha_full, params_kw, total_experts, rank, world, group, maps, timers):
    """policy "balanced": contribute_balanced, one all-reduce, device-side pick."""
    n_total = int(ha_full.shape[0])
    ex = _exchange(engine.device, n_total, world)
    buf, nxt = ex.take()
    contribute_balanced(engine, scene_coords, ha_full, params_kw, total_experts, rank, world, maps, timers, buf=buf)
    if dist.is_initialized():  # (also in a one-rank group: the collective's dtype / stream path is then the one N ranks take)
        _all_reduce_sum(buf, group, timers, engine)  # the one collective of this path
    return _pick_and_clear(ex, buf, nxt, n_total, world, engine)


def _pick_and_clear(ex, buf, nxt, n_total, world, engine):
    """The winner pick that ends a call of the alternating pair; its launch clears `nxt` -- also when it refuses to declare a
    winner (TeamTimeout: the kernel has run)."""
    try:
        out = pick_global(buf, n_total, world, engine, zero=nxt)
    except TeamTimeout:
        if buf.is_cuda:
            ex.cleared()
        raise
    if buf.is_cuda:
        ex.cleared()
    else:
        nxt.zero_()
        ex.cleared()
    return out


def contribute_balanced(engine, scene_coords, ha_full, params_kw, total_experts, rank, world, maps="full", timers=None, buf=None):
    """Rank `rank`'s part of the exchange under policy "balanced": device-built shard (one launch), forward launches
    writing scores by GLOBAL index and the record into this rank's slot of the exchange buffer (returned).
    Per frame: the shard kernel and the forward chain -- no host round trip, no allocation, no scatter; `buf`: a buffer the
    caller knows to be zero outside this rank's slots (forward_sharded's alternating pair), else a persistent one that is
    cleared here first.
    maps="owned": `scene_coords` holds only the maps of this rank's expert range [first, last] =
    params_kw["expert_range"] (plan_balanced on the histogram), `total_experts` = E."""
    dev = engine.device
    n_total = int(ha_full.shape[0])
    params_kw = dict(params_kw)
    expert_range = params_kw.pop("expert_range", None)
    E_local, _, H, W = scene_coords.shape
    owned = maps == "owned"
    if owned:
        if expert_range is None or total_experts is None:
            raise ValueError('maps="owned" needs params_kw["expert_range"] = (first, last) and params_kw["total_experts"]')
        first, last = int(expert_range[0]), int(expert_range[1])
        if last >= first and E_local != last - first + 1:
            raise ValueError("scene_coords must hold the maps of experts %d..%d" % (first, last))
        E_total, base = int(total_experts), first
    else:
        E_total, base = int(E_local), 0
    ha_dev = ha_full if ha_full.is_cuda else ha_full.to(dev)
    if ha_dev.stride(0) != 1:
        ha_dev = ha_dev.contiguous()  # stride-0 expand() of --expertselection
    lo, hi = shard_range(n_total, rank, world)
    n_local = hi - lo
    key = (str(dev), n_total, world, rank)
    ws = _balanced_ws.get(key)
    if ws is None:
        ws = _balanced_ws[key] = (torch.empty(max(n_local, 1), dtype=torch.int32, device=dev),
                                 torch.empty(max(n_local, 1), dtype=torch.int64, device=dev),
                                 torch.empty(4, dtype=torch.int32, device=dev))
    if buf is None:
        buf = _exchange_buffer(dev, n_total, world)
        buf.zero_()
    if n_local > 0:
        ev = None
        if timers is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        gidx, ha_local, _ = engine.shard_balanced(ha_dev, world, rank, E_total, expert_base=base, index_out=ws[0], assign_out=ws[1],
                                                  info_out=ws[2])
        if ev is not None:
            ev[1].record()
            timers.append(("shard", ev))
        p = _shard_params(engine, E_local, H, W, n_local, 0, dict(params_kw, scores_by_index=True, expert_base=base))
        engine.set_hyp_index(p, gidx)
        rec0 = n_total + rank * RES_DOUBLES
        engine.forward_device(scene_coords, ha_local, p, scores_out=buf[:n_total], result_out=buf[rec0:rec0 + RES_DOUBLES], want_host=False)
    return buf
