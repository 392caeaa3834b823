"""GPU parity of the training path: esac_hip_backward (through the C ABI / `esac.backward`) vs the CPU oracle's
restatement of esac_backward (esac.cpp:213-520), itself pinned bit-for-bit to the reference sources by
tests/test_oracle_vs_ref.py.  Same inputs, same Philox key.

Bars: index work (sampled cells, the set of hypotheses with p >= PROB_THRESH, accepted refinement steps, inlier
counts) bit-exact; probabilities, losses, poses and gradients to the floating-point tolerances written below
(the reference's own arithmetic is fp64 with float stores; the kernels reduce in a different, fixed order).
"""
import numpy as np
import pytest
import torch

from esac_amd import api
from esac_amd import synthetic as S

pytestmark = pytest.mark.gpu

# float32 gradient tensor, relative to the largest entry of the reference gradient:
GRAD_RTOL = 2e-6          # every cell that no selected hypothesis sampled (float rounding of the += chain is ~1e-7)
GRAD_RTOL_SAMPLED = 1e-3  # the 4 cells of each minimal set also receive d pose / d point from CENTRAL DIFFERENCES of the
                          # P3P solver with a step of 1e-3 (esac_derivative.h:128-185): (f - b) / 2e-3 multiplies the
                          # solver's own repeatability (poses agree to 1e-6 between the two implementations, see
                          # test_gpu_parity) by 500 -- the reference's value is that noisy by construction
LOSS_RTOL = 1e-7


def _gt(frame, rng=None, noise=0.0):
    """float32 4x4 ground-truth camera pose, optionally perturbed so that loss/dLoss are not at their minimum."""
    gt = np.array(frame["gt_pose"], np.float64)
    if noise:
        gt[:3, 3] += rng.normal(size=3) * noise
    return gt.astype(np.float32)


def _run_both(engine, oracle, frame, ha, gt, seed=1305, call=0, alpha=100.0, w_rot=1.0, w_trans=100.0, cut=100.0,
              want_paths=False, **kw):
    sc = torch.from_numpy(frame["coords"]).cuda()
    hat = torch.from_numpy(ha).cuda()
    E, _, H, W = frame["coords"].shape
    p = engine.make_params(E, H, W, len(ha), shift_x=frame["shift"][0], shift_y=frame["shift"][1], focal=frame["focal"],
                           ppx=frame["ppx"], ppy=frame["ppy"], sub_sampling=frame["sub"], inlier_alpha=alpha, seed=seed,
                           call=call, **kw)
    g_dev = torch.zeros_like(sc)
    out = engine.backward_device(sc, g_dev, hat, gt, w_rot, w_trans, cut, p)
    g_ref = np.zeros_like(frame["coords"])
    ref = oracle.backward(frame["coords"], g_ref, ha, gt, w_rot=w_rot, w_trans=w_trans, loss_cut=cut,
                          shift_x=frame["shift"][0], shift_y=frame["shift"][1], focal=frame["focal"], ppx=frame["ppx"],
                          ppy=frame["ppy"], sub_sampling=frame["sub"], inlier_alpha=alpha, seed=seed, call=call,
                          max_tries=kw.get("max_tries", 0), max_ref_steps=kw.get("max_ref_steps", -1), want_paths=want_paths)
    return out, g_dev.cpu().numpy(), ref, g_ref


def _check(engine, out, g_dev, ref, g_ref, expect_slots=None):
    N = len(ref["probs"])
    # stage: sampling (index work) and initial hypotheses
    np.testing.assert_array_equal(engine.read(api.BUF_SAMPLE_XY), ref["sample_xy"])
    np.testing.assert_allclose(engine.read(api.BUF_HYPS), ref["init_hyps"], rtol=0, atol=1e-6)
    # stage: distribution
    probs = engine.read(api.BUF_BWD_PROBS)
    np.testing.assert_allclose(probs, ref["probs"], rtol=1e-8, atol=1e-14)
    assert abs(out[2] - ref["entropy"]) < 1e-9
    sel_ref = np.nonzero(ref["probs"] >= 1e-3)[0]
    n_sel = int(out[1])
    # a hypothesis whose probability sits within rounding of the threshold may legitimately fall either side
    edge = np.abs(ref["probs"] - 1e-3) < 1e-12
    if not edge.any():
        assert n_sel == len(sel_ref)
        np.testing.assert_array_equal(engine.read(api.BUF_BWD_SLOTS)[:n_sel], sel_ref)  # ordered, ascending
    if expect_slots is not None:
        assert n_sel >= expect_slots, n_sel
    # stage: refinement of every selected hypothesis, loss, dLoss, score gradients
    np.testing.assert_allclose(engine.read(api.BUF_BWD_REF_HYPS), ref["ref_hyps"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(engine.read(api.BUF_BWD_LOSSES), ref["losses"], rtol=1e-6, atol=1e-6)
    # p_h (loss_h - E[loss]): the refined poses agree with the oracle's to ~1e-9 (the rounding of the LM sums through the
    # damped normal equations), the losses weigh translations by w_trans = 100, and the difference of two losses loses a
    # digit or two: 2e-5 relative
    np.testing.assert_allclose(engine.read(api.BUF_BWD_SCORE_GRADS), ref["score_grads"], rtol=2e-5, atol=1e-9)
    dl = engine.read(api.BUF_BWD_DLOSS)[:n_sel]
    if not edge.any():
        ref_dl = ref["dloss"][sel_ref]
        np.testing.assert_allclose(dl, ref_dl, rtol=1e-5, atol=1e-6 * max(1.0, np.abs(ref_dl).max()))
    # the value esac.backward returns
    assert abs(out[0] - ref["loss"]) <= LOSS_RTOL * max(1.0, abs(ref["loss"])), (out[0], ref["loss"])
    # the gradient tensor
    scale = max(float(np.abs(g_ref).max()), 1e-30)
    E, _, H, W = g_ref.shape
    sampled = np.zeros((E, H, W), bool)
    for h in sel_ref:
        for x, y in ref["sample_xy"][h]:
            sampled[:, y, x] = True
    diff = np.abs(g_dev - g_ref)
    err = float(diff[:, :, ~sampled.any(0)].max()) / scale
    err_s = float(diff[:, :, sampled.any(0)].max()) / scale
    print("backward parity: slots=%d loss_err=%.2e grad_err=%.2e (sampled cells %.2e), max |g| = %.3g"
          % (n_sel, abs(out[0] - ref["loss"]), err, err_s, scale))
    assert err <= GRAD_RTOL, (err, scale)
    assert err_s <= GRAD_RTOL_SAMPLED, (err_s, scale)
    assert np.isfinite(g_dev).all()
    return n_sel, err


@pytest.mark.parametrize("k", range(4))
def test_backward_config1_sharp_distribution(engine, oracle, k):
    """alpha = 100 (the training default): the softmax concentrates on a handful of hypotheses."""
    f = S.make_frame(k)
    ha = S.gating_assignment(f, 64 if k % 2 else 256)
    gt = _gt(f, np.random.default_rng(k), noise=0.02)
    out, g, ref, g_ref = _run_both(engine, oracle, f, ha, gt, call=k)
    n_sel, err = _check(engine, out, g, ref, g_ref, expect_slots=1)
    assert np.abs(g_ref).max() > 0


@pytest.mark.parametrize("k", range(3))
def test_backward_flat_distribution_many_slots(engine, oracle, k):
    """small alpha: dozens of hypotheses pass PROB_THRESH, every slab and the ordered float accumulation are exercised."""
    f = S.make_frame(10 + k)
    ha = S.gating_assignment(f, 128)
    gt = _gt(f, np.random.default_rng(k), noise=0.05)
    out, g, ref, g_ref = _run_both(engine, oracle, f, ha, gt, call=k, alpha=2.0)
    n_sel, err = _check(engine, out, g, ref, g_ref, expect_slots=20)


def test_backward_soft_clamped_loss(engine, oracle):
    """lossCut below the loss: sqrt(cut*loss) in the loss, the reference's 0.5/sqrt(loss) factor in dLoss."""
    f = S.make_frame(21)
    ha = S.gating_assignment(f, 64)
    gt = _gt(f, np.random.default_rng(3), noise=0.3)
    out, g, ref, g_ref = _run_both(engine, oracle, f, ha, gt, call=5, alpha=5.0, cut=2.0)
    assert (ref["losses"] > 2.0).any()
    _check(engine, out, g, ref, g_ref, expect_slots=2)


def test_backward_multi_expert_gating(engine, oracle):
    """3 experts, hypotheses spread by the gating distribution: gradients land in the right expert's map."""
    f = S.make_frame(31, E=3, true_expert=1)
    ha = S.gating_assignment(f, 96, mode="gating")
    gt = _gt(f, np.random.default_rng(4), noise=0.02)
    out, g, ref, g_ref = _run_both(engine, oracle, f, ha, gt, call=2, alpha=10.0)
    _check(engine, out, g, ref, g_ref, expect_slots=1)
    for e in range(3):
        assert (np.abs(g[e]).max() > 0) == (np.abs(g_ref[e]).max() > 0)


def test_backward_odd_grid_and_shift(engine, oracle):
    """W % 4 != 0 (scalar error pass), shifted crop (padX/padY of train_esac.py:159-160)."""
    f = S.make_frame(41, H=45, W=61, sub=10, shift=(7, -5))
    ha = S.gating_assignment(f, 64)
    gt = _gt(f, np.random.default_rng(5), noise=0.02)
    out, g, ref, g_ref = _run_both(engine, oracle, f, ha, gt, call=1, alpha=20.0)
    _check(engine, out, g, ref, g_ref, expect_slots=1)


def test_backward_path_slabs(engine, oracle):
    """The two gradient paths separately (oracle stage outputs) for the most probable hypothesis."""
    f = S.make_frame(51)
    ha = S.gating_assignment(f, 64)
    gt = _gt(f, np.random.default_rng(6), noise=0.05)
    out, g, ref, g_ref = _run_both(engine, oracle, f, ha, gt, call=3, alpha=3.0, want_paths=True)
    _check(engine, out, g, ref, g_ref, expect_slots=5)
    # path I is non-zero for a refined hypothesis, path II for every selected one
    sel = np.nonzero(ref["probs"] >= 1e-3)[0]
    assert any(np.abs(ref["grad_path1"][h]).max() > 0 for h in sel)
    assert all(np.abs(ref["grad_path2"][h]).max() > 0 for h in sel)


def test_backward_accumulates_into_existing_gradients(engine, oracle):
    """outGradients is `+=`-ed (esac.cpp:491-508): a second call with the same key doubles the first."""
    f = S.make_frame(61)
    ha = S.gating_assignment(f, 64)
    gt = _gt(f, np.random.default_rng(7), noise=0.05)
    sc = torch.from_numpy(f["coords"]).cuda()
    hat = torch.from_numpy(ha).cuda()
    E, _, H, W = f["coords"].shape
    p = engine.make_params(E, H, W, 64, focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"], inlier_alpha=5.0,
                           seed=9, call=4)
    g = torch.zeros_like(sc)
    l1 = engine.backward_device(sc, g, hat, gt, 1.0, 100.0, 100.0, p)
    g1 = g.clone()
    l2 = engine.backward_device(sc, g, hat, gt, 1.0, 100.0, 100.0, p)
    assert l1[0] == l2[0]  # deterministic
    g_ref1 = np.zeros_like(f["coords"])
    oracle.backward(f["coords"], g_ref1, ha, gt, focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"],
                    inlier_alpha=5.0, seed=9, call=4)
    g_ref2 = g_ref1.copy()
    oracle.backward(f["coords"], g_ref2, ha, gt, focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"],
                    inlier_alpha=5.0, seed=9, call=4)
    scale = np.abs(g_ref2).max()
    assert np.abs(g.cpu().numpy() - g_ref2).max() <= GRAD_RTOL_SAMPLED * scale
    assert np.abs(g1.cpu().numpy() - g_ref1).max() <= GRAD_RTOL_SAMPLED * scale
    np.testing.assert_allclose(g.cpu().numpy(), 2 * g1.cpu().numpy(), rtol=1e-6, atol=1e-7 * scale)


def test_backward_drop_in_call_with_cpu_tensors(engine, oracle):
    """`esac.backward` exactly as train_esac.py:151-168 calls it: CPU tensors in, gradients filled in place."""
    import esac
    f = S.make_frame(71)
    ha = S.gating_assignment(f, 64)
    gt = torch.from_numpy(_gt(f, np.random.default_rng(8), noise=0.05))
    pred = torch.from_numpy(f["coords"])
    grads = torch.zeros(pred.size())
    esac.set_seed(77, 3)
    loss = esac.backward(pred, grads, torch.from_numpy(ha), gt, 1.0, 100.0, 100.0, 0, 0, f["focal"], f["ppx"], f["ppy"],
                         10.0, 100.0, 0.5, 100.0, f["sub"])
    g_ref = np.zeros_like(f["coords"])
    ref = oracle.backward(f["coords"], g_ref, ha, gt.numpy(), focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"],
                          sub_sampling=f["sub"], seed=77, call=3)
    assert isinstance(loss, float)
    assert abs(loss - ref["loss"]) <= LOSS_RTOL * max(1.0, abs(ref["loss"]))
    assert np.abs(grads.numpy() - g_ref).max() <= GRAD_RTOL_SAMPLED * max(np.abs(g_ref).max(), 1e-30)
    assert esac.get_rng_state()[1] == 4  # one call consumed


def test_backward_matches_forward_hypotheses(engine, oracle):
    """A backward call draws the hypotheses of the forward call with the same (seed, call)."""
    f = S.make_frame(81)
    ha = S.gating_assignment(f, 64)
    sc = torch.from_numpy(f["coords"]).cuda()
    hat = torch.from_numpy(ha).cuda()
    E, _, H, W = f["coords"].shape
    p = engine.make_params(E, H, W, 64, focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"], seed=5, call=6)
    res = engine.forward_device(sc, hat, p)
    hyps_f = engine.read(api.BUF_HYPS).copy()
    engine.backward_device(sc, torch.zeros_like(sc), hat, _gt(f), 1.0, 100.0, 100.0, p)
    np.testing.assert_array_equal(engine.read(api.BUF_HYPS), hyps_f)
    # the forward winner is the most probable hypothesis of the backward distribution, refined to the same pose
    probs = engine.read(api.BUF_BWD_PROBS)
    w = int(res[api.RES_HYP])
    assert int(np.argmax(probs)) == w
    # (the forward's winner is refined by a team, the backward's slots by one workgroup each: the same re-fits with another
    # summation order of the LM sums)
    np.testing.assert_allclose(engine.read(api.BUF_BWD_REF_HYPS)[w], res[api.RES_RVEC:api.RES_RVEC + 6], rtol=0, atol=1e-8)


def _fd_check(engine, oracle, max_ref_steps, rel_bar, n_cells=3):
    f = S.make_frame(91)
    ha = S.gating_assignment(f, 32)
    gt = _gt(f, np.random.default_rng(9), noise=0.05)
    E, _, H, W = f["coords"].shape
    sc = torch.from_numpy(f["coords"]).cuda()
    hat = torch.from_numpy(ha).cuda()
    p = engine.make_params(E, H, W, 32, focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"], inlier_alpha=3.0,
                           seed=3, call=1, max_ref_steps=max_ref_steps)
    g = torch.zeros_like(sc)
    out = engine.backward_device(sc, g, hat, gt, 1.0, 100.0, 100.0, p)
    g = g.cpu().numpy()
    sampled = set(map(tuple, engine.read(api.BUF_SAMPLE_XY).reshape(-1, 2)))
    checked = 0
    for idx in np.argsort(-np.abs(g[0]).reshape(-1)):
        c, rem = divmod(int(idx), H * W)
        y, x = divmod(rem, W)
        if (x, y) in sampled:
            continue  # sampled cells move the hypotheses themselves (and can flip the accepted sampling try)
        h = 2e-3
        vals = []
        for s in (+1, -1):
            pert = f["coords"].copy()
            pert[0, c, y, x] += s * h
            vals.append(engine.backward_device(torch.from_numpy(pert).cuda(), torch.zeros_like(sc), hat, gt, 1.0, 100.0, 100.0, p)[0])
        fd = (vals[0] - vals[1]) / (2 * h)
        a = float(g[0, c, y, x])
        assert np.sign(fd) == np.sign(a), (fd, a)
        assert abs(fd - a) <= rel_bar * abs(a), (fd, a)
        checked += 1
        if checked == n_cells:
            break
    assert checked == n_cells


def test_backward_finite_difference_score_path(engine, oracle):
    """max_ref_steps = 0: no re-fit, so the expected loss depends on a non-sampled coordinate only through the scores
    (softmax derivative x sigmoid derivative x d error / d point) -- an exact derivative, finite differences of the
    returned loss must reproduce the gradient entry."""
    _fd_check(engine, oracle, max_ref_steps=0, rel_bar=0.02)


def test_backward_finite_difference_with_refinement(engine, oracle):
    """With refinement the reference's path I is a Gauss-Newton style approximation on the residual NORMS
    (esac.cpp:417-434: only the radial component of each 2-D residual enters), so it cannot match finite
    differences closely; sign and order of magnitude must still agree."""
    _fd_check(engine, oracle, max_ref_steps=-1, rel_bar=3.0)


# ---------------------------------------------------------------- edge cases
def test_backward_single_hypothesis(engine, oracle):
    """N = 1: probability 1, one slot."""
    f = S.make_frame(101)
    ha = S.gating_assignment(f, 1)
    out, g, ref, g_ref = _run_both(engine, oracle, f, ha, _gt(f, np.random.default_rng(1), noise=0.05), call=0)
    n_sel, _ = _check(engine, out, g, ref, g_ref, expect_slots=1)
    assert n_sel == 1 and ref["probs"][0] == 1.0


def test_backward_no_hypothesis_reaches_the_threshold(engine, oracle):
    """N = 2048 with a flat distribution: every p = 1/2048 < PROB_THRESH, nothing is refined, the gradient stays as
    it was, the loss is the plain mean of the initial hypotheses' losses."""
    f = S.make_frame(102, H=24, W=32, sub=20)
    ha = S.gating_assignment(f, 2048)
    gt = _gt(f, np.random.default_rng(2), noise=0.05)
    out, g, ref, g_ref = _run_both(engine, oracle, f, ha, gt, call=1, alpha=1e-4)
    assert int(out[1]) == 0 and (ref["probs"] < 1e-3).all()
    assert not g.any() and not g_ref.any()
    assert abs(out[0] - ref["loss"]) <= LOSS_RTOL * abs(ref["loss"])
    np.testing.assert_allclose(engine.read(api.BUF_BWD_REF_HYPS), ref["init_hyps"], rtol=0, atol=1e-6)


def test_backward_more_hypotheses_than_slots_can_exist(engine, oracle):
    """N = 1500 > 1000: the slot list is bounded by the threshold itself; a moderately flat distribution selects a subset."""
    f = S.make_frame(103, H=24, W=32, sub=20)
    ha = S.gating_assignment(f, 1500)
    gt = _gt(f, np.random.default_rng(3), noise=0.05)
    out, g, ref, g_ref = _run_both(engine, oracle, f, ha, gt, call=2, alpha=4.0)
    n_sel, _ = _check(engine, out, g, ref, g_ref, expect_slots=10)
    assert n_sel < 1000


def test_backward_sampling_budget_exhausted(engine, oracle):
    """A map of pure noise and 3 tries: most hypotheses keep the state of their last try (or the zero pose of a failed
    solve, esac_util.h:107-111); losses, probabilities and gradients still follow the reference."""
    f = S.make_frame(104, outlier_frac=1.0)
    ha = S.gating_assignment(f, 48)
    gt = _gt(f, np.random.default_rng(4), noise=0.05)
    out, g, ref, g_ref = _run_both(engine, oracle, f, ha, gt, call=3, alpha=5.0, max_tries=3)
    assert (engine.read(api.BUF_TRIES) == -1).any()
    _check(engine, out, g, ref, g_ref)


def test_backward_grid_larger_than_the_lds_list(engine, oracle):
    """P = 100 x 120 = 12000 cells > 8192: the slot refinement keeps its correspondence lists in global memory."""
    f = S.make_frame(105, H=100, W=120, sub=4)
    ha = S.gating_assignment(f, 24)
    gt = _gt(f, np.random.default_rng(5), noise=0.05)
    out, g, ref, g_ref = _run_both(engine, oracle, f, ha, gt, call=4, alpha=3.0)
    _check(engine, out, g, ref, g_ref, expect_slots=4)


def test_backward_rejects_sharded_calls_and_bad_pointers(engine):
    f = S.make_frame(106)
    sc = torch.from_numpy(f["coords"]).cuda()
    ha = torch.zeros(16, dtype=torch.int64, device="cuda")
    p = engine.make_params(1, 60, 80, 16, hyp_offset=16)
    with pytest.raises(RuntimeError, match="sharded"):
        engine.backward_device(sc, torch.zeros_like(sc), ha, np.eye(4, dtype=np.float32), 1.0, 100.0, 100.0, p)
    p = engine.make_params(1, 60, 80, 16)
    with pytest.raises(RuntimeError, match="singular"):
        engine.backward_device(sc, torch.zeros_like(sc), ha, np.zeros((4, 4), np.float32), 1.0, 100.0, 100.0, p)
    with pytest.raises(RuntimeError):
        engine.backward_device(sc, torch.zeros(1, 3, 60, 79, device="cuda"), ha, np.eye(4, dtype=np.float32), 1.0, 100.0, 100.0, p)


def test_slot_teams_equal_one_workgroup_per_slot(oracle, monkeypatch):
    """The training path refines its selected hypotheses ("slots") by TEAMS of 8 workgroups on one XCD each when THE CALL ITSELF
    selects few enough (<= 32, decided on the device: both launches are issued, one of them returns at once), one workgroup
    per slot otherwise -- the route is a function of the call's inputs, so the first call on a context already takes it and
    an identical second call returns the same bits.  Same frame, same key: expected loss, slot list, refined poses and the
    gradient tensor of both routes against each other and against the oracle; then a team that never completes
    (ESAC_DEBUG_COOP_STALL): the call refines again with one workgroup per slot, returns that route's result, and teams stay
    off on that context until esac_hip_set_refine_team re-arms them."""
    monkeypatch.setenv("ESAC_SLOT_TEAMS", "0")
    solo = api.Engine(0)  # a context that never asks for slot teams
    monkeypatch.delenv("ESAC_SLOT_TEAMS")
    eng = api.Engine(0)   # a context of its own: the shared one may have latched
    f = S.make_frame(83)
    ha = S.gating_assignment(f, 96)
    gt = _gt(f)
    _run = lambda e, o, fr, a, seed, call: _run_both(e, o, fr, a, gt, seed=seed, call=call)
    first = _run(solo, oracle, f, ha, 7, 3)
    assert not solo.bwd_team_info()["teams"] and solo.bwd_team_info()["slots"] == int(first[0][1]) <= 32
    _check(solo, *first)
    solo_refs = solo.read(api.BUF_BWD_REF_HYPS).copy()
    second = _run(eng, oracle, f, ha, 7, 3)  # the FIRST call on this context: teams at once
    info = eng.bwd_team_info()
    assert info["teams"] and info["team_calls"] == 1 and info["team_fallbacks"] == 0, info
    _check(eng, *second)
    assert abs(second[0][0] - first[0][0]) <= 1e-9 * max(1.0, abs(first[0][0]))
    np.testing.assert_allclose(eng.read(api.BUF_BWD_REF_HYPS), solo_refs, rtol=0, atol=1e-8)
    scale = np.abs(first[1]).max()
    assert np.abs(second[1] - first[1]).max() <= 1e-3 * scale  # (sampled cells: the finite-difference path amplifies 1e-9 pose differences)
    again = _run(eng, oracle, f, ha, 7, 3)
    np.testing.assert_array_equal(again[1], second[1])  # same inputs, same key, same route: same bits, whatever ran before
    # a member that never shows up
    eng.set_debug(coop_stall=True)
    try:
        third = _run(eng, oracle, f, ha, 7, 3)
    finally:
        eng.set_debug()
    info = eng.bwd_team_info()
    assert not info["teams"] and info["team_fallbacks"] == 1, info
    _check(eng, *third)
    np.testing.assert_array_equal(third[1], first[1])  # one workgroup per slot: bit for bit the other context's result
    fourth = _run(eng, oracle, f, ha, 7, 3)
    assert not eng.bwd_team_info()["teams"] and eng.bwd_team_info()["team_calls"] == 3  # (the stalled call counted, no team call since)
    np.testing.assert_array_equal(fourth[1], first[1])
    eng.set_refine_team(8)  # an explicit request re-arms the teams
    fifth = _run(eng, oracle, f, ha, 7, 3)
    assert eng.bwd_team_info()["teams"]
    np.testing.assert_array_equal(fifth[1], second[1])
