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


@pytest.mark.parametrize("native", [True, False])
@pytest.mark.parametrize("policy", ["range", "expert", "balanced"])
def test_forward_sharded_world1_matches_plain(engine, policy, native, monkeypatch):
    """native: the collective is the LIBRARY's own ncclAllReduce (esac_hip_comm_init / esac_hip_allreduce_sum, id bootstrapped over
    the torch.distributed group); else torch.distributed.all_reduce on the same RCCL."""
    import torch.distributed as dist
    monkeypatch.setenv("ESAC_NATIVE_RCCL", "1" if native else "0")
    if not native:
        engine.comm_destroy()
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
        timers = []
        scores_g, best = D.forward_sharded(engine, sc, hat, kw, policy=policy, timers=timers)
        # the ONE collective has run -- a one-rank RCCL all-reduce(SUM) of the real float64 exchange buffer on the launch stream
        assert [n for n, _ in timers].count("allreduce") == 1
        assert (engine._comm == (1, 0)) == native
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


@pytest.mark.parametrize("N,E,world", [(250, 3, 2), (4096, 12, 4), (16384, 50, 8), (5, 2, 8), (1000, 1, 3), (777, 4096, 5)])
def test_device_built_shard_equals_the_host_plan(engine, N, E, world):
    """esac_hip_shard_balanced (one launch, no host round trip) against its CPU mirror: every rank's index SET, its
    local assignment, its expert range and n_local -- incl. more ranks than hypotheses, a single expert split by index,
    the 4096-expert limit and out-of-range values (counted as expert 0, copied unchanged)."""
    rng = np.random.default_rng(N + E)
    p = rng.dirichlet(np.full(E, 0.3)) if E > 1 else np.ones(1)
    ha = rng.choice(E, size=N, p=p).astype(np.int64)
    if N > 100:
        ha[17], ha[N // 2] = -3, E + 5
    hat = torch.from_numpy(ha).cuda()
    seen = []
    for rank in range(world):
        ref_idx, ref_rng = D.shard_balanced_host(ha, rank, world, E=E)
        for base_mode in (0, 1):
            base = ref_rng[0] if base_mode and len(ref_idx) else 0
            gidx, ha_local, info = engine.shard_balanced(hat, world, rank, E, expert_base=base)
            torch.cuda.synchronize()
            g, hl, info = gidx.cpu().numpy(), ha_local.cpu().numpy(), info.cpu().numpy()
            assert len(g) == len(ref_idx) == info[2]
            assert sorted(g.tolist()) == sorted(ref_idx.tolist())
            raw = ha[g]
            ok = (raw >= 0) & (raw < E)
            np.testing.assert_array_equal(hl[ok], raw[ok] - base)
            np.testing.assert_array_equal(hl[~ok], raw[~ok])       # out-of-range values travel unchanged
            if len(g):
                assert (int(info[0]), int(info[1])) == ref_rng
                e_clamped = np.where(ok, raw, 0)
                assert (np.diff(e_clamped) >= 0).all()             # sorted by expert inside the shard
            assert int(info[3]) == int(((ha < 0) | (ha >= E)).any())
        seen.append(g)
    assert sorted(np.concatenate(seen).tolist()) == list(range(N))


@pytest.mark.parametrize("world", [2, 4])
def test_balanced_contributions_of_several_ranks_on_one_device(engine, world):
    """The per-rank half of the balanced exchange for every rank of a `world`-rank job, one after the other on this box's
    single GPU (summing the buffers is what the all-reduce does), with the real per-frame path: device-built shard, scores
    written by global index, record with the global expert id -- full maps and owned expert ranges."""
    E, N = 6, 1500
    f = S.make_frame(63, E=E, true_expert=4)
    ha = S.gating_assignment(f, N, mode="dirichlet")
    ha[::7] = 4
    sc_full = torch.from_numpy(f["coords"]).cuda()
    hat = torch.from_numpy(ha).cuda()
    kw = dict(seed=1305, call=11)
    res = engine.forward_device(sc_full, hat, engine.make_params(E, 60, 80, N, **kw))
    flags = engine.read(api.BUF_EXACT_FLAGS).astype(bool)
    full = engine.read(api.BUF_SCORES)
    plan = D.plan_balanced(np.bincount(ha, minlength=E), world)
    for maps in ("full", "owned"):
        total = torch.zeros(N + world * D.RES_DOUBLES, dtype=torch.float64, device="cuda")
        for rank in range(world):
            pk = dict(kw)
            sc = sc_full
            if maps == "owned":
                first, last = plan[rank]
                sc = sc_full[first:last + 1].contiguous()
                pk.update(total_experts=E, expert_range=(first, last))
            total_experts = pk.pop("total_experts", None)
            total += D.contribute_balanced(engine, sc, hat, pk, total_experts, rank, world, maps)
        scores_g, best = D.pick_global(total, N, world)
        assert int(best[api.RES_HYP]) == int(res[api.RES_HYP]) and int(best[api.RES_EXPERT]) == int(res[api.RES_EXPERT]), maps
        np.testing.assert_array_equal(best[api.RES_POSE:api.RES_POSE + 16], res[api.RES_POSE:api.RES_POSE + 16])
        assert best[api.RES_SCORE] == res[api.RES_SCORE]
        got = scores_g.cpu().numpy()
        np.testing.assert_array_equal(got[flags], full[flags])
        np.testing.assert_allclose(got[~flags], full[~flags], rtol=0, atol=2e-3)


def _two_rank_worker(rank, world, port, policy, q):
    """One of `world` processes sharing this box's single GPU: own engine / context, gloo group (RCCL refuses two
    ranks on one device), the real forward_sharded."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        eng = api.Engine(0)
        f = S.make_frame(62, E=3, true_expert=1)
        ha = torch.from_numpy(S.gating_assignment(f, 250, mode="gating")).cuda()
        maps = "full"
        extra = {}
        if policy == "expert-owned":  # this rank holds ONLY its experts' maps (e % world == rank)
            policy, maps = "expert", "owned"
            sc = torch.from_numpy(np.ascontiguousarray(f["coords"][D.owned_experts(3, rank, world)])).cuda()
        elif policy == "balanced-owned":  # ... only the maps of its expert range under the balanced plan
            policy, maps = "balanced", "owned"
            first, last = D.plan_balanced(np.bincount(ha.cpu().numpy(), minlength=3), world)[rank]
            sc = torch.from_numpy(np.ascontiguousarray(f["coords"][first:last + 1])).cuda()
            extra = dict(expert_range=(first, last))
        else:
            sc = torch.from_numpy(f["coords"]).cuda()
        out = []
        for call in (20, 21):  # two frames: the persistent exchange buffer is re-zeroed between calls
            kw = dict(seed=1305, call=call)
            if maps == "owned":
                kw["total_experts"] = 3
            kw.update(extra)
            scores_g, best = D.forward_sharded(eng, sc, ha, kw, policy=policy, maps=maps)
            out.append((scores_g.cpu().numpy().copy(), best.copy()))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,policy", [(2, "range"), (3, "range"), (2, "expert"), (2, "expert-owned"), (3, "expert-owned"),
                                          (2, "balanced"), (3, "balanced"), (2, "balanced-owned"), (3, "balanced-owned")])
def test_forward_sharded_across_processes_on_one_device(engine, world, policy):
    """The whole multi-rank path as the bench drives it -- forward_sharded in `world` processes, one collective per
    frame -- with gloo standing in for RCCL: every rank ends with the same winner, and it is the unsharded winner."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, policy, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    f = S.make_frame(62, E=3, true_expert=1)
    ha = torch.from_numpy(S.gating_assignment(f, 250, mode="gating")).cuda()
    sc = torch.from_numpy(f["coords"]).cuda()
    for i, call in enumerate((20, 21)):
        res = engine.forward_device(sc, ha, engine.make_params(3, 60, 80, 250, seed=1305, call=call))
        for rank in range(world):
            scores_g, best = results[rank][i]
            assert int(best[api.RES_HYP]) == int(res[api.RES_HYP]) and int(best[api.RES_EXPERT]) == int(res[api.RES_EXPERT])
            np.testing.assert_array_equal(best[api.RES_POSE:api.RES_POSE + 16], res[api.RES_POSE:api.RES_POSE + 16])
            assert best[api.RES_SCORE] == res[api.RES_SCORE]
            np.testing.assert_array_equal(scores_g, results[0][i][0])  # every rank holds the same global score vector


def _stalled_rank_worker(rank, world, port, q):
    """forward_sharded in `world` processes sharing the GPU (gloo group); rank 1's refinement teams never fill
    (ESAC_DEBUG_COOP_STALL): its record travels as ESAC_RES_VALID = 3, the pick refuses on every rank, every rank runs the frame
    again with ESAC_FLAG_REFINE_SOLO."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        eng = api.Engine(0)
        if rank == 1:
            eng.set_debug(coop_stall=True)
        f = S.make_frame(64)
        ha = torch.from_numpy(S.gating_assignment(f, 256)).cuda()
        sc = torch.from_numpy(f["coords"]).cuda()
        out = []
        for call in (30, 31, 32):
            scores_g, best = D.forward_sharded(eng, sc, ha, dict(seed=1305, call=call), policy="range")
            out.append((best.copy(), eng.refine_info()))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_team_timeout_of_one_rank_is_retried_by_every_rank(engine):
    """The advisor's round-5 finding: at several ranks the forward calls are asynchronous, so a team time-out on one rank used to
    leave that rank's record without ESAC_RES_VALID and the winner came silently from the others.  Now the pick refuses (-12)
    on every rank alike and all of them repeat the frame with one workgroup per refinement; the failing rank's context counts
    the strikes and latches after two."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stalled_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    f = S.make_frame(64)
    ha = torch.from_numpy(S.gating_assignment(f, 256)).cuda()
    sc = torch.from_numpy(f["coords"]).cuda()
    engine.set_refine_team(0)
    try:
        for i, call in enumerate((30, 31, 32)):
            res = engine.forward_device(sc, ha, engine.make_params(1, 60, 80, 256, seed=1305, call=call))
            for rank in range(2):
                best, info = results[rank][i]
                assert int(best[api.RES_HYP]) == int(res[api.RES_HYP]), (rank, call)
                assert best[api.RES_SCORE] == res[api.RES_SCORE]
                np.testing.assert_allclose(best[api.RES_POSE:api.RES_POSE + 16], res[api.RES_POSE:api.RES_POSE + 16], rtol=0, atol=1e-6)
    finally:
        engine.set_refine_team(api.REFINE_TEAM_DEFAULT)
    info1 = [info for _, info in results[1]]
    assert info1[0]["team_fallbacks"] == 1 and not info1[0]["team_latched_off"], info1
    assert info1[1]["team_fallbacks"] == 2 and info1[1]["team_latched_off"], info1
    assert info1[2]["team_fallbacks"] == 2 and info1[2]["mode"] == "one_workgroup" and not info1[2]["timed_out"], info1  # latched: no wait
    assert all(info["team_fallbacks"] == 0 for _, info in results[0])  # the healthy rank only repeats the frame


def test_failed_record_is_not_an_empty_shard(engine):
    """k_pick_record / pick_global: a record with ESAC_RES_VALID = 3 (team time-out) stops the pick (TeamTimeout) -- on the
    device and in the host scan of a CPU buffer; the same frame with ESAC_FLAG_REFINE_SOLO on every rank gives the plain call's
    winner."""
    f = S.make_frame(65)
    ha = torch.from_numpy(S.gating_assignment(f, 256)).cuda()
    sc = torch.from_numpy(f["coords"]).cuda()
    kw = dict(seed=1305, call=40)
    world, n_total = 2, 256

    def exchange(params_kw, stall_rank=None):
        total = torch.zeros(n_total + world * D.RES_DOUBLES, dtype=torch.float64, device="cuda")
        for rank in range(world):
            engine.set_debug(coop_stall=rank == stall_rank)
            buf = torch.empty_like(total)
            D.contribute_range(engine, sc, ha, params_kw, rank, world, buf)
            total += buf
        engine.set_debug()
        return total

    try:
        total = exchange(kw, stall_rank=1)
        assert float(total[n_total + D.RES_DOUBLES + D.RES_VALID]) == 3.0 and float(total[n_total + D.RES_VALID]) == 1.0
        with pytest.raises(D.TeamTimeout):
            D.pick_global(total, n_total, world, engine)
        with pytest.raises(D.TeamTimeout):
            D.pick_global(total.cpu(), n_total, world)
        total = exchange(dict(kw, refine_solo=True), stall_rank=1)  # the stall cannot bite: no team is asked for
        _, best = D.pick_global(total, n_total, world, engine)
        assert engine.refine_info()["mode"] == "one_workgroup"
    finally:
        engine.set_debug()
        engine.set_refine_team(api.REFINE_TEAM_DEFAULT)
    res = engine.forward_device(sc, ha, engine.make_params(1, 60, 80, n_total, **kw))
    # (the winner's exact score: the shards re-scored it in the selection kernel, the plain call in its team's prologue, member by
    # member -- the same terms in another order)
    assert int(best[api.RES_HYP]) == int(res[api.RES_HYP]) and abs(best[api.RES_SCORE] - res[api.RES_SCORE]) <= 1e-12 * abs(res[api.RES_SCORE])
    np.testing.assert_allclose(best[api.RES_POSE:api.RES_POSE + 16], res[api.RES_POSE:api.RES_POSE + 16], rtol=0, atol=1e-6)


def test_exchange_pair_survives_a_call_that_ended_early(engine):
    """_Exchange: a call that raises between take() and its pick leaves the OTHER buffer holding the all-reduced data of the call
    before; the next taker clears it (dirty flag) instead of summing stale scores and records into the new frame."""
    ex = D._Exchange(torch.device("cuda", 0), 8, 2)
    cur, nxt = ex.take()
    cur.fill_(5.0)          # call i: its data, all-reduced
    nxt.zero_(); ex.cleared()  # ... and its pick cleared the buffer of call i + 1
    cur2, nxt2 = ex.take()  # call i + 1 starts (clean buffer) ...
    assert cur2.data_ptr() == nxt.data_ptr() and float(cur2.abs().sum()) == 0.0
    cur2.fill_(7.0)         # ... and dies before its pick: nxt2 (= call i's buffer) was never cleared
    cur3, _ = ex.take()     # call i + 2 gets call i's buffer: cleared on take
    assert cur3.data_ptr() == cur.data_ptr() and float(cur3.abs().sum()) == 0.0


def _run_bench(world, extra):
    import json
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ESAC_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(world)] + extra,
                         capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_multi_rank_path_on_one_device():
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one rank per GPU), here with
    two ranks sharing the one GPU through the ESAC_BENCH_ONE_DEVICE test hook (gloo instead of RCCL): barrier-bracketed
    timing, max over ranks, ONE JSON line from rank 0 with the whole-job aggregate."""
    d = _run_bench(2, ["--steps", "30", "--warmup", "4"])
    assert d["n_gpus"] == 2 and d["steps"] == 30 and d["warmup"] == 4 and d["scaling"] == "weak"
    assert d["config"]["hypotheses_total"] == 512 and d["config"]["shard_sizes"] == [256, 256] and d["config"]["policy"] == "range"
    assert d["value"] > 0 and abs(d["value"] - 512 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "batched" not in d and "cpu_baseline" not in d  # single-GPU extras only
    assert d["allreduce_ms"] is not None and d["allreduce_ms"] > 0
    assert d["ranks_seen"] == 2 and d["devices"] == [0, 0] and d["launcher"] == "external"


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (how the driver's scaling sweep may call it): bench.py starts the two
    ranks itself and relays rank 0's one JSON line -- n_gpus is the number of ranks that ran, ranks_seen what their exchange's
    communicator reported, devices the GPU of every rank.  More GPUs than the node has: a non-zero exit, never a one-GPU line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ESAC_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["devices"] == [0, 0] and len(d["rank_reports"]) == 2
    assert sorted(r["rank"] for r in d["rank_reports"]) == [0, 1] and len({r["pid"] for r in d["rank_reports"]}) == 2
    assert "bench.py itself" in d["launcher"] and d["config"]["hypotheses_total"] == 512
    env.pop("ESAC_BENCH_ONE_DEVICE")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "9", "--steps", "6", "--warmup", "2"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert out.returncode != 0 and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "--gpus 9" in out.stderr


@pytest.mark.parametrize("world", [2, 3])
def test_bench_expert_sharded_strong_scaling_on_one_device(world):
    """BASELINE configs[3] as the driver would run it on several GPUs: the load-balanced split (every rank N / world
    hypotheses of a contiguous expert range, holding only those experts' maps, shard built on the device inside the
    step), strong scaling -- here 2 and 3 ranks on the one GPU."""
    d = _run_bench(world, ["--config", "cfg4", "--steps", "6", "--warmup", "2"])
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["config"]["policy"] == "balanced"
    sizes = d["config"]["shard_sizes"]
    assert d["config"]["hypotheses_total"] == 4096 and sum(sizes) == 4096 and len(sizes) == world
    assert max(sizes) / (sum(sizes) / world) <= 1.10
    assert d["value"] > 0 and abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "only the maps of its own expert range" in d["config"]["parallelism"]
    assert d["shard_build_ms"] is not None and d["shard_build_ms"] > 0  # built per frame, inside the timed step
