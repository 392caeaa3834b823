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
