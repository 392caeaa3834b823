"""The sharded forward on a real GPU through RCCL (backend "nccl") with the world size this box offers (1):
exercises esac_amd/distributed.py end to end on device tensors -- global-index shards, in-kernel record,
pack, all_reduce, pick -- and must equal the plain call.  (Rank-count independence itself is covered by the
gloo tests on the CPU and test_sharding_independence_on_device.)"""
import os
import socket

import numpy as np
import pytest
import torch

from esac_amd import api
from esac_amd import distributed as D
from esac_amd import synthetic as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("policy", ["range", "expert"])
def test_forward_sharded_world1_matches_plain(engine, policy):
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        f = S.make_frame(60, E=3, true_expert=2)
        ha = S.gating_assignment(f, 192, mode="gating")
        sc = torch.from_numpy(f["coords"]).cuda()
        hat = torch.from_numpy(ha).cuda()
        kw = dict(seed=1305, call=8)
        scores_g, best = D.forward_sharded(engine, sc, hat, kw, policy=policy)
        p = engine.make_params(3, 60, 80, 192, **kw)
        res = engine.forward_device(sc, hat, p)
        assert int(best[api.RES_HYP]) == int(res[api.RES_HYP]) and int(best[api.RES_EXPERT]) == int(res[api.RES_EXPERT])
        np.testing.assert_array_equal(best[api.RES_POSE:api.RES_POSE + 16], res[api.RES_POSE:api.RES_POSE + 16])
        np.testing.assert_array_equal(scores_g.cpu().numpy(), engine.read(api.BUF_SCORES))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_range_contributions_of_several_ranks_on_one_device(engine, world):
    """The per-rank half of the exchange for every rank of a `world`-rank job, run one after the other on this
    box's single GPU; summing the buffers is what the all-reduce does.  Winner, pose and the score of every
    hypothesis the unsharded call scored exactly must equal the unsharded call."""
    f = S.make_frame(61, E=2, true_expert=1)
    ha = S.gating_assignment(f, 250, mode="gating")  # 250: ragged shards for world = 3 and 8
    sc = torch.from_numpy(f["coords"]).cuda()
    hat = torch.from_numpy(ha).cuda()
    kw = dict(seed=1305, call=9)
    n_total = 250
    total = torch.zeros(n_total + world * D.RES_DOUBLES, dtype=torch.float64, device="cuda")
    for rank in range(world):
        buf = torch.empty_like(total)
        D.contribute_range(engine, sc, hat, kw, rank, world, buf)
        total += buf
    scores_g, best = D.pick_global(total, n_total, world)
    p = engine.make_params(2, 60, 80, n_total, **kw)
    res = engine.forward_device(sc, hat, p)
    assert int(best[api.RES_HYP]) == int(res[api.RES_HYP]) and int(best[api.RES_EXPERT]) == int(res[api.RES_EXPERT])
    np.testing.assert_array_equal(best[api.RES_POSE:api.RES_POSE + 16], res[api.RES_POSE:api.RES_POSE + 16])
    assert best[api.RES_SCORE] == res[api.RES_SCORE]
    flags = engine.read(api.BUF_EXACT_FLAGS).astype(bool)
    full = engine.read(api.BUF_SCORES)
    got = scores_g.cpu().numpy()
    # within the margin of the global maximum = within the margin of its shard's maximum: exact on both sides
    np.testing.assert_array_equal(got[flags], full[flags])
    # elsewhere a shard may have re-scored exactly what the full call left at its fp32 score
    np.testing.assert_allclose(got[~flags], full[~flags], rtol=0, atol=2e-3)
