"""World-size-2 (and 3) gloo tests of the multi-GPU exchange (esac_amd/distributed.py) on the CPU.

Each rank evaluates ITS shard with the oracle standing in for the HIP engine (the exchange logic under test
is engine-agnostic), contributes one zero-padded buffer to ONE all-reduce(SUM), and every rank must end up
with exactly the single-process result: same winner (global index), same expert, same pose, same score vector.
This is the property that makes the 1/2/4/8-GPU runs comparable (results independent of the rank count)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from esac_amd import distributed as D
from esac_amd import synthetic as S


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_local(O, f, ha_full, gidx, call):
    """Local result record (same layout as the HIP result record) + local scores for shard `gidx`."""
    rec = np.zeros(D.RES_DOUBLES)
    if len(gidx) == 0:
        return np.zeros(0), rec
    o = O.forward(f["coords"], ha_full[gidx], seed=1305, call=call, hyp_index=gidx.astype(np.int32), num_threads=1)
    w = o["winner"]
    rec[0] = o["scores"][w]
    rec[1] = gidx[w]
    rec[2] = o["expert"]
    rec[3:9] = o["refined"]
    rec[9:25] = o["pose"].reshape(-1)
    rec[25] = o["ref_steps"]
    return o["scores"], rec


class _Untouchable:
    """Stands for an Engine in the gloo workers: any use of it is a test failure."""

    def __getattr__(self, name):
        raise AssertionError("the engine was used on a gloo group: %s" % name)


def _worker(rank, world, port, policy, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import esac_oracle as O
    try:
        f = S.make_frame(21, E=4, true_expert=2)
        ha = S.gating_assignment(f, 96, mode="dirichlet")
        n_total = len(ha)
        if policy == "range":
            lo, hi = D.shard_range(n_total, rank, world)
            gidx = np.arange(lo, hi)
        elif policy == "balanced":
            gidx = D.shard_balanced_host(ha, rank, world, E=4)[0].astype(np.int64)
        else:
            gidx = D.shard_by_expert(ha, rank, world).astype(np.int64)
        scores_l, rec_l = _oracle_local(O, f, ha, gidx, call=4)
        buf = D.pack_local(torch.from_numpy(np.ascontiguousarray(scores_l)), torch.from_numpy(rec_l), n_total,
                           torch.from_numpy(gidx.astype(np.int32)), rank, world)
        assert buf.numel() == n_total + world * D.RES_DOUBLES
        # THE one collective, through the product's own entry: on a gloo group (and for a CPU buffer) the library's RCCL
        # communicator must stay out of it -- the engine is never touched
        assert D.native_comm(_Untouchable(), None) is None
        D._all_reduce_sum(buf, None, engine=_Untouchable())
        scores_g, best = D.pick_global(buf, n_total, world)
        q.put((rank, scores_g.numpy().copy(), best.copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,policy", [(2, "range"), (2, "expert"), (3, "range"), (2, "balanced"), (3, "balanced")])
def test_sharded_forward_matches_single_process(oracle, world, policy):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, policy, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    f = S.make_frame(21, E=4, true_expert=2)
    ha = S.gating_assignment(f, 96, mode="dirichlet")
    ref = oracle.forward(f["coords"], ha, seed=1305, call=4, num_threads=1)
    for rank, scores_g, best in results:
        np.testing.assert_array_equal(scores_g, ref["scores"])          # identical score vector on every rank
        assert int(best[1]) == ref["winner"] and int(best[2]) == ref["expert"]
        np.testing.assert_array_equal(best[9:25].reshape(4, 4).astype(np.float32), ref["pose"])
        assert best[0] == ref["scores"][ref["winner"]]


def test_shard_helpers():
    for n, w in [(256, 1), (256, 8), (10, 3), (5, 8)]:
        spans = [D.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    ha = np.array([0, 3, 1, 2, 3, 3, 0, 1])
    parts = [D.shard_by_expert(ha, r, 2) for r in range(2)]
    assert sorted(np.concatenate(parts).tolist()) == list(range(8))
    assert all((ha[p] % 2 == r).all() for r, p in enumerate(parts))


def _bench_assignments(cfg):
    """The assignment vectors bench.py draws for a preset (same generators, same frames)."""
    E, N, mode = {"cfg4": (12, 4096, "gating"), "cfg5a": (50, 16384, "dirichlet")}[cfg]
    out = []
    for k in range(3):
        f = S.make_frame(k, E=E, H=6, W=8)  # the gating draw does not depend on the grid
        out.append((E, S.gating_assignment(f, N, mode=mode)))
    return out


@pytest.mark.parametrize("cfg,world", [("cfg4", 4), ("cfg5a", 8), ("cfg4", 3), ("cfg5a", 5)])
def test_balanced_policy_balances_where_expert_ownership_does_not(cfg, world):
    """"a load-balanced assignment from the hypAssignment histogram".  On the bench's own generators the
    e % world split puts most hypotheses on one rank (cfg4 on 4 ranks: ~3990 of 4096); the balanced split gives every
    rank N / world (+-1), is a partition of the hypotheses, keeps a rank's experts contiguous and shares only the
    experts at the cuts."""
    for E, ha in _bench_assignments(cfg):
        n = len(ha)
        parts = [D.shard_balanced_host(ha, r, world, E=E) for r in range(world)]
        sizes = np.array([len(idx) for idx, _ in parts])
        assert sizes.max() / sizes.mean() <= 1.10 and sizes.max() - sizes.min() <= 1
        assert sorted(np.concatenate([idx for idx, _ in parts]).tolist()) == list(range(n))
        plan = D.plan_balanced(np.bincount(ha, minlength=E), world)
        for r, (idx, rng) in enumerate(parts):
            assert rng == plan[r]                                   # the histogram alone gives every rank's expert range
            assert ha[idx].min() == rng[0] and ha[idx].max() == rng[1]
            assert (np.diff(ha[idx]) >= 0).all()                    # (expert, index) order
            if r:
                assert plan[r][0] >= plan[r - 1][1]                 # ranges only touch at the cut experts
        naive = np.array([len(D.shard_by_expert(ha, r, world)) for r in range(world)])
        assert naive.max() / naive.mean() > 1.10                    # what the e % world split does on the same vector


def test_plan_balanced_edge_cases():
    assert D.plan_balanced([0, 5, 0], 2) == [(1, 1), (1, 1)]            # one expert, split by index
    assert D.plan_balanced([2, 0, 2], 2) == [(0, 0), (2, 2)]            # empty experts in between are nobody's
    assert D.plan_balanced([1, 1], 4) == [(0, 0), (1, 1), (0, -1), (0, -1)]  # more ranks than hypotheses
    idx, rng = D.shard_balanced_host(np.array([7, -1, 1, 1]), 0, 2, E=2)  # out-of-range values count as expert 0
    assert idx.tolist() == [0, 1] and rng == (0, 0)


def test_pick_global_tie_rule_and_empty_shards():
    """Ties go to the lowest GLOBAL hypothesis index (esac_util.h:519 first-max); empty shards are ignored."""
    n_total, world = 6, 3
    recs = np.zeros((world, D.RES_DOUBLES))
    recs[0, :3] = [50.0, 4, 1]
    recs[0, -1] = 1
    recs[1, :3] = [50.0, 2, 0]
    recs[1, -1] = 1  # same score, lower index -> wins
    buf = torch.from_numpy(np.concatenate([np.arange(n_total, dtype=np.float64), recs.reshape(-1)]))
    _, best = D.pick_global(buf, n_total, world)
    assert int(best[1]) == 2 and int(best[2]) == 0
    with pytest.raises(RuntimeError):
        D.pick_global(torch.zeros(n_total + world * D.RES_DOUBLES, dtype=torch.float64), n_total, world)
