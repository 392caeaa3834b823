"""The speculative forward route (round 6; esac_capi.hip: forward_impl, esac_kernels.hip: k_spec_join): with several experts the
sampler's straggler chain -- the hypotheses its first pass of 32 tries did not settle -- runs on a stream of the context's own
BESIDE the score / selection / refinement of the settled hypotheses (esac_util.h:152-223 is the per-hypothesis retry loop,
esac.cpp:167-177 refines the winner only); a join kernel completes the selection over all hypotheses and delivers, or finds that
the winner is not the hypothesis that was refined and has the refinement run again.

The bar: EVERY output -- sampled cells, accepted tries, poses, score vector, exact flags, statistics, refinement trace, record --
is bit for bit what the serial route (ESAC_DEBUG_NO_SPECULATION) produces, whatever the workspace held before; and the serial
route is what the other parity tests hold against the oracle.  The speculative call always runs in ANOTHER order than the serial
reference calls, so that what it finds in the workspace is some other frame's state, never its own serial twin's.
"""
import numpy as np
import pytest
import torch

from esac_amd import api
from esac_amd import synthetic as S

pytestmark = pytest.mark.gpu

KEYS = ("tries", "xy", "hyps", "flags", "scores", "user", "counts", "imap")


def _run(engine, frame, ha, call, nospec, seed=1305, want_device_record=False, second_best=False):
    E, _, H, W = frame["coords"].shape
    sc, hat = torch.from_numpy(frame["coords"]).cuda(), torch.from_numpy(ha).cuda()
    engine.set_debug(no_speculation=nospec, spec_second_best=second_best)
    try:
        p = engine.make_params(E, H, W, len(ha), focal=frame["focal"], ppx=frame["ppx"], ppy=frame["ppy"], sub_sampling=frame["sub"],
                               seed=seed, call=call)
        scores = torch.full((len(ha),), -7.0, dtype=torch.float64, device="cuda")
        dev_rec = torch.full((32,), -7.0, dtype=torch.float64, device="cuda") if want_device_record else None
        rec = engine.forward_device(sc, hat, p, scores_out=scores, result_out=dev_rec, want_host=not want_device_record)
        torch.cuda.synchronize()
        out = dict(rec=None if rec is None else rec.copy(), hyps=engine.read(api.BUF_HYPS), tries=engine.read(api.BUF_TRIES),
                   xy=engine.read(api.BUF_SAMPLE_XY), scores=engine.read(api.BUF_SCORES), flags=engine.read(api.BUF_EXACT_FLAGS),
                   user=scores.cpu().numpy(), counts=engine.read(api.BUF_INLIER_COUNTS), imap=engine.read(api.BUF_INLIER_MAP),
                   info=engine.spec_info(), result=engine.read(api.BUF_RESULT))
        if want_device_record:
            out["dev_rec"] = dev_rec.cpu().numpy()
        if not nospec:
            out["stragglers"] = engine.read(api.BUF_SPEC_FLAGS).astype(bool)
        return out
    finally:
        engine.set_debug()


def _assert_same(a, b, what):
    for key in KEYS:
        np.testing.assert_array_equal(a[key], b[key], err_msg="%s: %s" % (what, key))
    np.testing.assert_array_equal(a["result"][:31], b["result"][:31], err_msg="%s: workspace record" % what)
    if a["rec"] is not None and b["rec"] is not None:
        np.testing.assert_array_equal(a["rec"], b["rec"], err_msg="%s: host record" % what)


@pytest.mark.parametrize("E,N,mode", [(3, 300, "gating"), (10, 1024, "gating"), (12, 4096, "gating"), (6, 1500, "dirichlet"), (4, 8192, "gating")])
def test_speculative_route_equals_the_serial_route(engine, E, N, mode):
    """BASELINE configs[2] / [3] shapes and their neighbours (the latency sampler with hand-over at N <= 1024, the throughput
    sampler's first pass above; the 8192-hypothesis limit of the route)."""
    frames = {k: S.make_frame(300 + k, E=E) for k in range(5)}
    has = {k: S.gating_assignment(frames[k], N, mode=mode) for k in frames}
    serial = {k: _run(engine, frames[k], has[k], call=k, nospec=True) for k in frames}
    held = failed = 0
    for k in sorted(frames, reverse=True):
        spec = _run(engine, frames[k], has[k], call=k, nospec=False)
        assert not serial[k]["info"]["last_speculative"] and spec["info"]["last_speculative"]
        assert spec["stragglers"].any() and not spec["stragglers"].all()
        # a straggler is a hypothesis the first pass (32 tries; 128 at N <= 256) did not settle
        t = spec["tries"][spec["stragglers"]]
        assert ((t >= 32) | (t == -1)).all()
        _assert_same(serial[k], spec, "E=%d N=%d frame %d" % (E, N, k))
        failed += spec["info"]["last_failed"]
        held += not spec["info"]["last_failed"]
        # the speculation fails exactly when the winner is a straggler, or a straggler moved the band past the speculative winner
        if spec["stragglers"][int(spec["rec"][api.RES_HYP])]:
            assert spec["info"]["last_failed"]
    assert held + failed == len(frames)


def test_a_straggler_that_wins(engine, oracle):
    """Constructed: the true expert's map with 60 % outliers -- a hypothesis on it needs ~40 tries for four inlying cells, so about
    half of them are stragglers of the first pass, and they score as well as the settled ones: in some frames the winner IS a
    straggler.  The join then has the refinement run again for it and the call returns what the serial route returns; held
    against the oracle as well (winner, accepted tries, refinement trace, pose)."""
    won = held = 0
    for k in range(10):
        f = S.make_frame(500 + k, E=2, true_expert=1, outlier_frac=0.6)
        ha = S.gating_assignment(f, 512, mode="gating")
        serial = _run(engine, f, ha, call=40 + k, nospec=True)
        other = S.make_frame(900 + k, E=2)  # another frame's state in the workspace in between
        _run(engine, other, S.gating_assignment(other, 512, mode="gating"), call=k, nospec=False)
        spec = _run(engine, f, ha, call=40 + k, nospec=False)
        _assert_same(serial, spec, "frame %d" % k)
        winner_is_straggler = bool(spec["stragglers"][int(spec["rec"][api.RES_HYP])])
        won += winner_is_straggler
        held += not spec["info"]["last_failed"]
        if winner_is_straggler:
            assert spec["info"]["last_failed"]
            if won == 1:
                ref = oracle.forward(f["coords"], ha, focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"], seed=1305, call=40 + k)
                assert int(spec["rec"][api.RES_HYP]) == ref["winner"] and int(spec["rec"][api.RES_EXPERT]) == ref["expert"]
                np.testing.assert_array_equal(spec["tries"], ref["tries"])
                np.testing.assert_array_equal(spec["counts"], ref["inlier_counts"])
                np.testing.assert_array_equal(spec["imap"], ref["inlier_map"])
                r_err, t_err = S.pose_errors(spec["rec"][api.RES_POSE:api.RES_POSE + 16].reshape(4, 4), ref["pose"])
                assert r_err <= 1e-4 and t_err <= 1e-3
    assert won >= 1 and held >= 1, (won, held)  # both ways out of the join were taken


@pytest.mark.parametrize("E,N", [(3, 300), (10, 1024), (12, 4096)])
def test_the_refined_hypothesis_is_not_the_winner_of_the_settled(engine, E, N):
    """The speculative refinement starts from the fp32 argmax of the settled hypotheses while the selection proper (band, exact
    re-scores) runs beside it; the two can disagree when the fp32 stream and the reference arithmetic order two near-equal scores
    differently.  ESAC_DEBUG_SPEC_SECOND_BEST forces that situation (the refinement starts from the runner-up): the join must find
    that the refined hypothesis is not the winner, the gated second refinement must run, and every output -- blocking and
    asynchronous -- is the serial route's."""
    for k in range(3):
        f = S.make_frame(1200 + k, E=E)
        ha = S.gating_assignment(f, N, mode="gating")
        serial = _run(engine, f, ha, call=20 + k, nospec=True)
        other = S.make_frame(1300 + k, E=E)
        _run(engine, other, S.gating_assignment(other, N, mode="gating"), call=k, nospec=False)
        spec = _run(engine, f, ha, call=20 + k, nospec=False, second_best=True)
        assert spec["info"]["last_speculative"] and spec["info"]["last_failed"]
        _assert_same(serial, spec, "second best, frame %d" % k)
        _run(engine, other, S.gating_assignment(other, N, mode="gating"), call=k, nospec=False)
        asyn = _run(engine, f, ha, call=20 + k, nospec=False, second_best=True, want_device_record=True)
        assert asyn["info"]["last_failed"] and asyn["dev_rec"][31] == 1.0
        for key in KEYS:
            np.testing.assert_array_equal(serial[key], asyn[key], err_msg=key)
        np.testing.assert_array_equal(asyn["dev_rec"][:31], serial["rec"][:31])
        engine.check()


def test_nothing_settled_by_the_first_pass(engine):
    """85 % outliers on the true expert's map and garbage on the other: (nearly) every hypothesis is a straggler.  With NO settled
    contender the speculative refinement has nothing to refine and says so (hypothesis -1 in its record); the join sends the call
    to the second refinement."""
    f = S.make_frame(610, E=2, true_expert=0, outlier_frac=0.85)
    ha = S.gating_assignment(f, 300, mode="gating")
    serial = _run(engine, f, ha, call=3, nospec=True)
    g = S.make_frame(611, E=2)
    _run(engine, g, S.gating_assignment(g, 300, mode="gating"), call=4, nospec=False)
    spec = _run(engine, f, ha, call=3, nospec=False)
    _assert_same(serial, spec, "all stragglers")
    settled_contenders = (~spec["stragglers"]) & spec["flags"].astype(bool)
    if not settled_contenders.any():
        assert spec["info"]["last_failed"]
    assert spec["stragglers"].mean() > 0.9


def test_asynchronous_call_delivers_the_same_device_record(engine):
    """No host record (the multi-GPU exchange's calls): the second refinement is enqueued with the call and runs only when the
    join marked the speculation as failed; the device record (ESAC_RES_VALID = 1) and the score vector are the blocking call's."""
    seen_failed = seen_held = False
    for k in range(8):
        f = S.make_frame(700 + k, E=2, true_expert=1, outlier_frac=0.6)
        ha = S.gating_assignment(f, 512, mode="gating")
        blocking = _run(engine, f, ha, call=60 + k, nospec=True)
        h = S.make_frame(800 + k, E=2)
        _run(engine, h, S.gating_assignment(h, 512, mode="gating"), call=k, nospec=False)
        asyn = _run(engine, f, ha, call=60 + k, nospec=False, want_device_record=True)
        for key in KEYS:
            np.testing.assert_array_equal(blocking[key], asyn[key], err_msg=key)
        assert asyn["dev_rec"][31] == 1.0
        np.testing.assert_array_equal(asyn["dev_rec"][:31], blocking["rec"][:31])
        engine.check()
        seen_failed |= asyn["info"]["last_failed"]
        seen_held |= not asyn["info"]["last_failed"]
    assert seen_failed and seen_held


def test_where_the_route_applies(engine):
    """Several experts, a single frame, the screened sampler, the fp32 ranking stream, 256 < N <= 8192: speculative.  One expert,
    exact routes, the selection folded into the refinement kernel (N <= 256), more than 8192 hypotheses: serial."""
    def last(E, N, **kw):
        f = S.make_frame(650, E=E)
        ha = S.gating_assignment(f, N, mode="gating")
        sc, hat = torch.from_numpy(f["coords"]).cuda(), torch.from_numpy(ha).cuda()
        engine.forward_device(sc, hat, engine.make_params(E, 60, 80, N, seed=1305, call=1, **kw))
        return engine.spec_info()["last_speculative"]
    before = engine.spec_info()["calls"]
    assert last(3, 512) and last(12, 4096)
    assert engine.spec_info()["calls"] == before + 2
    assert not last(1, 512)
    assert not last(3, 192)                      # the team's prologue selects: another summation order of the exact re-scores
    assert not last(3, 512, exact_scores=True)
    assert not last(3, 512, exact_sampling=True)
    assert not last(50, 16384)                   # the chain takes every hypothesis from try 0: nothing is settled early
    engine.set_debug(no_speculation=True)
    try:
        assert not last(3, 512)
    finally:
        engine.set_debug()
    assert last(3, 512)


def test_streams_that_share_one_hardware_queue():
    """The route's three streams hand over through polled words, and every waiter is enqueued behind the launch it waits for -- so
    streams that the runtime maps onto ONE hardware queue (GPU_MAX_HW_QUEUES=1: everything serialises) cannot wait for each other:
    the same comparisons in a process of that kind -- no time-out (status 5 would switch the speculation off and the
    `last_speculative` assertions would fail), every output the serial route's."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_speculation.py"), "-q", "-x", "-m", "gpu", "-k",
                        "equals_the_serial_route or straggler_that_wins or not_the_winner_of_the_settled or asynchronous"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:]


def test_a_chain_that_never_reports_is_a_bounded_wait_and_a_serial_rerun(oracle):
    """ESAC_DEBUG_SPEC_LOSE_CHAIN leaves out the launch that writes "the chain is done" (as if it had failed, or something else held
    its queue).  Every wait of the route is bounded: the join gives up after 20 ms and reports status 5, the gated second refinement
    (whose patience outlasts the join's) closes, the blocking call runs again in stream order and returns THAT route's outputs, and
    the context stops speculating; an asynchronous call's device record says "not a record" (ESAC_RES_VALID = 3)."""
    import time
    eng = api.Engine(0)  # a context of its own: this test switches its speculation off for good
    f = S.make_frame(1400, E=3)
    ha = S.gating_assignment(f, 512, mode="gating")
    serial = _run(eng, f, ha, call=9, nospec=True)
    spec = _run(eng, f, ha, call=9, nospec=False)
    assert spec["info"]["last_speculative"]
    _assert_same(serial, spec, "the route itself, before the chain is lost")
    # asynchronous: nobody reruns the call -- the record must not pass for one
    eng.set_debug(spec_lose_chain=True)
    try:
        E, _, H, W = f["coords"].shape
        sc, hat = torch.from_numpy(f["coords"]).cuda(), torch.from_numpy(ha).cuda()
        p = eng.make_params(E, H, W, len(ha), focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"], seed=1305, call=9)
        dev_rec = torch.full((32,), -7.0, dtype=torch.float64, device="cuda")
        t0 = time.perf_counter()
        eng.forward_device(sc, hat, p, result_out=dev_rec, want_host=False)
        torch.cuda.synchronize()
        dt_async = time.perf_counter() - t0
        assert dev_rec.cpu().numpy()[31] == 3.0
        assert 0.015 < dt_async < 0.5, dt_async  # the join's 20 ms, not a hang
        # blocking: status 5 -> the same call again, in stream order
        scores = torch.full((len(ha),), -7.0, dtype=torch.float64, device="cuda")
        t0 = time.perf_counter()
        rec = eng.forward_device(sc, hat, p, scores_out=scores)
        dt = time.perf_counter() - t0
        assert 0.015 < dt < 0.5, dt
        np.testing.assert_array_equal(rec, serial["rec"])
        np.testing.assert_array_equal(scores.cpu().numpy(), serial["user"])
        for key, buf in (("hyps", api.BUF_HYPS), ("tries", api.BUF_TRIES), ("scores", api.BUF_SCORES), ("flags", api.BUF_EXACT_FLAGS),
                         ("counts", api.BUF_INLIER_COUNTS), ("imap", api.BUF_INLIER_MAP)):
            np.testing.assert_array_equal(eng.read(buf), serial[key], err_msg=key)
        info = eng.spec_info()
        assert not info["last_speculative"]  # (the rerun was the stream-order route)
    finally:
        eng.set_debug()
    # ... and the context has stopped speculating: the next call is in stream order at once, and right
    again = _run(eng, f, ha, call=9, nospec=False)
    assert not again["info"]["last_speculative"]
    _assert_same(serial, again, "after the context stopped speculating")
    ref = oracle.forward(f["coords"], ha, focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"], seed=1305, call=9)
    assert int(again["rec"][api.RES_HYP]) == ref["winner"]
