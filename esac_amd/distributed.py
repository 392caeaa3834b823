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
depend on the number of ranks.  The payload is (N_total + 32*world) * 8 bytes
(<= 133 KB at N = 16384, 8 ranks): latency-bound, xGMI bandwidth is irrelevant.
"""
import numpy as np
import torch
import torch.distributed as dist

RES_SCORE, RES_HYP, RES_DOUBLES = 0, 1, 32


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


def pick_global(buf, n_total, world, engine=None):
    """(scores_global [n_total], winning record [32]) from the all-reduced buffer.  With a device buffer and an engine
    the pick runs on the device (esac_hip_pick_record: one launch, the record lands in pinned host memory); a CPU
    buffer (gloo tests) is scanned on the host."""
    scores = buf[:n_total]
    if engine is not None and buf.is_cuda:
        try:
            return scores, engine.pick_record(buf[n_total:], world)
        except RuntimeError as exc:
            if "no rank produced" in str(exc):
                raise RuntimeError("esac: no rank produced a hypothesis")
            raise
    recs = buf[n_total:].view(world, RES_DOUBLES).cpu().numpy()
    best = None
    for r in range(world):
        rec = recs[r]
        if rec[RES_DOUBLES - 1] != 1.0:
            continue  # empty shard
        if best is None or rec[RES_SCORE] > best[RES_SCORE] or (rec[RES_SCORE] == best[RES_SCORE] and rec[RES_HYP] < best[RES_HYP]):
            best = rec
    if best is None:
        raise RuntimeError("esac: no rank produced a hypothesis")
    return scores, best.copy()


_buffers = {}


def _exchange_buffer(dev, n_total, world):
    """Persistent all-reduce payload per (device, N, world): no allocation on the per-frame path."""
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


def contribute_range(engine, scene_coords, hyp_assign_full, params_kw, rank, world, buf):
    """Rank `rank`'s part of the exchange for contiguous-range sharding: zero `buf`, run the forward path on
    hypotheses [lo, hi) with the kernels writing scores and record into this rank's slots of `buf`."""
    n_total = int(hyp_assign_full.shape[0])
    E, _, H, W = scene_coords.shape
    lo, hi = shard_range(n_total, rank, world)
    buf.zero_()
    if hi > lo:
        dev = engine.device
        ha_dev = hyp_assign_full if hyp_assign_full.is_cuda else hyp_assign_full.to(dev)
        if ha_dev.stride(0) != 1:
            ha_dev = ha_dev.contiguous()  # stride-0 expand() of --expertselection
        p = _shard_params(engine, E, H, W, hi - lo, lo, params_kw)
        rec0 = n_total + rank * RES_DOUBLES
        # the refinement kernel writes the record itself, incl. the "this rank contributed" marker in its last slot
        # (ESAC_RES_VALID, the convention pack_local follows for the index-list path)
        engine.forward_device(scene_coords, ha_dev[lo:hi], p, scores_out=buf[lo:hi], result_out=buf[rec0:rec0 + RES_DOUBLES],
                              want_host=False)
    return buf


def _all_reduce_sum(buf, group, timers=None):
    """The one collective.  RCCL ("nccl") reduces the device buffer in place; a gloo group (CPU tests, or several
    ranks sharing one GPU) gets the 1-2 KB payload staged through the host.  `timers`: optional list that receives a
    (start, end) pair of CUDA events around the collective (bench.py splits its time out of the step)."""
    ev = None
    if timers is not None and buf.is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if buf.is_cuda and dist.get_backend(group) == "gloo":
        host = buf.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        buf.copy_(host)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    if ev is not None:
        ev[1].record()
        timers.append(ev)


def owned_experts(E, rank, world):
    """Experts whose maps rank `rank` holds under policy "expert": e % world == rank (local index e // world)."""
    return list(range(rank, E, world))


def forward_sharded(engine, scene_coords, hyp_assign_full, params_kw, group=None, policy="range", maps="full", timers=None):
    """Multi-GPU esac_forward: every rank holds `scene_coords` (or only its experts' maps) and the full assignment
    vector; returns (scores_global [N] f64 device tensor, winning record np[32]).

    `params_kw` are the keyword arguments of Engine.make_params except N / hyp_offset.
    policy "range": contiguous index ranges -- the kernels write this rank's scores and record straight into its
    slots of the persistent exchange buffer (hyp_offset keys RNG and tie-breaks), so a call is one memset, the
    forward launches, the all-reduce and the winner pick.
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
        # the returned score vector is a view of the persistent buffer: valid until the next call on this device
        buf = contribute_range(engine, scene_coords, ha_full, params_kw, rank, world, _exchange_buffer(dev, n_total, world))
        if world > 1:
            _all_reduce_sum(buf, group, timers)  # the one collective of this path
        return pick_global(buf, n_total, world, engine)
    if policy != "expert":
        raise ValueError(policy)
    owned = maps == "owned"
    if owned and total_experts is None:
        raise ValueError('maps="owned" needs params_kw["total_experts"]')
    key = ("expert", str(dev), n_total, world, rank, owned, int(ha_full.data_ptr()) if ha_full.is_cuda else id(ha_full))
    shard = _shard_cache.get(key)
    if shard is None:  # the shard of an assignment vector: index list + local assignment, built once per vector
        if len(_shard_cache) > 64:
            _shard_cache.clear()
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
    if world > 1:
        _all_reduce_sum(buf, group, timers)  # the one collective of this path
    return pick_global(buf, n_total, world, engine)


_shard_cache = {}
