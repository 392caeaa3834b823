"""GPU edge cases the reference's semantics imply (ragged widths, exhausted budgets, ties, strides, big grids)."""
import numpy as np
import pytest
import torch

from esac_amd import api
from esac_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _both(engine, oracle, coords, ha, **kw):
    E, _, H, W = coords.shape
    p = engine.make_params(E, H, W, len(ha), **kw)
    res = engine.forward_device(torch.from_numpy(np.ascontiguousarray(coords)).cuda(), torch.from_numpy(np.ascontiguousarray(ha)).cuda(), p)
    okw = {k: v for k, v in kw.items() if k in ("shift_x", "shift_y", "focal", "ppx", "ppy", "inlier_thresh", "inlier_alpha",
                                                "inlier_beta", "max_reproj", "sub_sampling", "seed", "call", "max_tries", "max_ref_steps")}
    ref = oracle.forward(coords, ha, **okw)
    return res, ref


def _same(engine, res, ref):
    np.testing.assert_array_equal(engine.read(api.BUF_TRIES), ref["tries"])
    np.testing.assert_array_equal(engine.read(api.BUF_SAMPLE_XY), ref["sample_xy"])
    assert int(res[api.RES_HYP]) == ref["winner"] and int(res[api.RES_EXPERT]) == ref["expert"]
    assert int(res[api.RES_REF_STEPS]) == ref["ref_steps"]
    np.testing.assert_array_equal(engine.read(api.BUF_INLIER_COUNTS), ref["inlier_counts"])
    np.testing.assert_array_equal(engine.read(api.BUF_INLIER_MAP), ref["inlier_map"])
    r_err, t_err = S.pose_errors(res[api.RES_POSE:api.RES_POSE + 16].reshape(4, 4), ref["pose"])
    assert r_err <= 1e-4 and t_err <= 1e-3, (r_err, t_err)


@pytest.mark.parametrize("H,W,sub", [(30, 41, 16), (24, 32, 20), (13, 17, 37), (120, 160, 4)])
def test_odd_grid_shapes(engine, oracle, H, W, sub):
    """W % 4 != 0 takes the scalar-load path of the score kernel; P not a multiple of the block sizes."""
    f = S.make_frame(30, H=H, W=W, sub=sub)
    ha = S.gating_assignment(f, 48)
    res, ref = _both(engine, oracle, f["coords"], ha, sub_sampling=sub, call=9)
    _same(engine, res, ref)


def test_shift_and_parameters(engine, oracle):
    f = S.make_frame(31, shift=(5, -3))
    ha = S.gating_assignment(f, 64)
    res, ref = _both(engine, oracle, f["coords"], ha, shift_x=5, shift_y=-3, inlier_thresh=6.0, inlier_alpha=50.0,
                     inlier_beta=0.8, max_reproj=60.0, call=2)
    _same(engine, res, ref)


def test_budget_exhaustion_constant_map(engine, oracle):
    """Every P3P fails on a constant map: zero pose kept, tries = -1 after max_tries (esac_util.h:107-111,154)."""
    coords = np.ones((1, 3, 12, 16), np.float32)
    ha = np.zeros(8, np.int64)
    res, ref = _both(engine, oracle, coords, ha, max_tries=130)  # not a multiple of 64: partial last round
    assert (engine.read(api.BUF_TRIES) == -1).all()
    np.testing.assert_array_equal(engine.read(api.BUF_HYPS), 0.0)
    np.testing.assert_array_equal(engine.read(api.BUF_SAMPLE_XY), ref["sample_xy"])  # state of the LAST try remains
    assert int(res[api.RES_HYP]) == ref["winner"] == 0
    assert int(res[api.RES_REF_STEPS]) == ref["ref_steps"]
    assert np.isfinite(res[api.RES_POSE:api.RES_POSE + 16]).all()


def test_small_try_budget_matches_oracle(engine, oracle):
    f = S.make_frame(32)
    ha = S.gating_assignment(f, 128)
    res, ref = _both(engine, oracle, f["coords"], ha, max_tries=3, call=1)  # many hypotheses run out of tries
    assert (ref["tries"] == -1).any() and (ref["tries"] >= 0).any()
    np.testing.assert_allclose(engine.read(api.BUF_HYPS), ref["hyps"], atol=1e-6)
    _same(engine, res, ref)


def test_tie_goes_to_first_index(engine):
    """Identical hypotheses -> identical exact scores -> argmax keeps the first (esac_util.h:519)."""
    f = S.make_frame(33)
    N = 40
    ha = S.gating_assignment(f, N)
    sc = torch.from_numpy(f["coords"]).cuda()
    hat = torch.from_numpy(ha).cuda()
    p = engine.make_params(1, 60, 80, N)
    engine.forward_device(sc, hat, p)
    hyps = engine.read(api.BUF_HYPS)
    best = int(np.argmax(engine.read(api.BUF_SCORES)))
    dup = np.tile(hyps[best], (N, 1))
    dup[:7] = hyps[(best + 1) % N]  # a different (worse or equal) hypothesis in front
    engine.write_hyps(dup)
    engine.score(sc, hat, p)
    engine.select(sc, hat, p)
    engine.refine(sc, hat, p)
    res = engine.read(api.BUF_RESULT)
    scores = engine.read(api.BUF_SCORES)
    assert len(set(scores[7:].tolist())) == 1
    first = 7 if scores[7] >= scores[0] else 0
    assert int(res[api.RES_HYP]) == first
    assert int(res[api.RES_CONTENDERS]) >= N - 7


@pytest.mark.parametrize("shape", ["stream", "tiled"])
def test_fast_score_of_a_hypothesis_far_beyond_the_scene(engine, shape):
    """The fp32 ranking stream forms err = d2n * rsq(d2n * zc^2): for a hypothesis whose camera sits ~1e10 m from the scene the
    product overflows fp32, rsq(inf) = 0, and every cell would read err = 0 -- a perfect inlier, the maximal score, for the
    worst hypothesis of the set.  It must read the clamped error maxReproj instead (score ~ 0), in both score shapes."""
    f = S.make_frame(35)
    N = 64
    ha = S.gating_assignment(f, N)
    sc, hat = torch.from_numpy(f["coords"]).cuda(), torch.from_numpy(ha).cuda()
    p = engine.make_params(1, 60, 80, N, score_shape=shape)
    engine.forward_device(sc, hat, p)
    hyps = engine.read(api.BUF_HYPS)
    good = float(np.max(engine.read(api.BUF_SCORES)))
    far = hyps.copy()
    far[5, 3:] = [1.0e9, -2.0e9, 1.0e10]   # zc ~ 1e10, d2n ~ (300 zc)^2: d2n zc^2 ~ 1e45
    far[9, 3:] = [0.0, 0.0, -3.0e10]
    engine.write_hyps(far)
    engine.score(sc, hat, p)
    engine.select(sc, hat, p)
    scores = engine.read(api.BUF_SCORES)
    assert scores[5] < 1e-3 * good and scores[9] < 1e-3 * good, (scores[5], scores[9], good)
    assert int(np.argmax(scores)) not in (5, 9)


def test_cpu_tensors_strides_and_expand(oracle):
    """Drop-in call with CPU tensors, a non-contiguous coordinate tensor and the stride-0 expand()
    assignment of --expertselection (test_esac.py:171-173)."""
    import esac
    f = S.make_frame(34, E=3, true_expert=1)
    big = torch.zeros(3, 3, 60, 160)
    big[..., ::2] = torch.from_numpy(f["coords"])
    sc = big[..., ::2]
    assert not sc.is_contiguous()
    ha = torch.tensor([1]).expand(64)
    assert ha.stride() == (0,)
    out = torch.zeros(4, 4)
    esac.set_seed(1305, 0)
    e = esac.forward(sc, ha, out, 0, 0, 525.0, 320.0, 240.0, 10.0, 100.0, 0.5, 100.0, 8)
    ref = oracle.forward(f["coords"], np.full(64, 1, np.int64))
    assert e == 1 == ref["expert"]
    r_err, t_err = S.pose_errors(out.numpy(), ref["pose"])
    assert r_err <= 1e-4 and t_err <= 1e-3
    with pytest.raises(RuntimeError):
        esac.forward(sc, torch.tensor([5]).expand(8), out, 0, 0, 525.0, 320.0, 240.0, 10.0, 100.0, 0.5, 100.0, 8)


def test_fast_score_band_is_wide_enough(engine, oracle):
    """|fp32 streaming score - exact score| must stay well inside the re-score band (alpha * 1e-3)."""
    worst = 0.0
    for k in range(6):
        f = S.make_frame(40 + k, E=2, true_expert=k % 2, grid_spacing=5.0)
        ha = S.gating_assignment(f, 256, mode="gating")
        sc = torch.from_numpy(f["coords"]).cuda()
        hat = torch.from_numpy(ha).cuda()
        p = engine.make_params(2, 60, 80, 256, call=k)
        engine.sample(sc, hat, p)
        engine.score(sc, hat, p)
        engine.select(sc, hat, p)
        fast = engine.read(api.BUF_SCORES).copy()
        flags = engine.read(api.BUF_EXACT_FLAGS).astype(bool)
        engine.score_exact(sc, hat, p)
        exact = engine.read(api.BUF_SCORES)
        worst = max(worst, np.abs(fast[~flags] - exact[~flags]).max())
    assert worst < 0.1 * 100.0 * 1e-3, worst  # 10x head-room inside the default band


def test_large_grid_global_list_path(engine, oracle):
    """P > 8192 cells: the refinement's correspondence list lives in global memory instead of LDS."""
    f = S.make_frame(50, H=120, W=160, sub=4)
    ha = S.gating_assignment(f, 32)
    res, ref = _both(engine, oracle, f["coords"], ha, sub_sampling=4, call=3)
    _same(engine, res, ref)


def test_sharding_independence_on_device(engine):
    """Shards evaluated with global hypothesis indices reproduce the unsharded stage outputs bit-exactly."""
    f = S.make_frame(51, E=3, true_expert=0)
    ha = S.gating_assignment(f, 96, mode="dirichlet")
    sc = torch.from_numpy(f["coords"]).cuda()
    p = engine.make_params(3, 60, 80, 96, call=6)
    engine.forward_device(sc, torch.from_numpy(ha).cuda(), p)
    xy_all, hyps_all = engine.read(api.BUF_SAMPLE_XY).copy(), engine.read(api.BUF_HYPS).copy()
    scores_all = engine.read(api.BUF_SCORES).copy()
    flags_all = engine.read(api.BUF_EXACT_FLAGS).astype(bool)
    for idx in (np.arange(40, 96), np.nonzero(ha % 2 == 1)[0]):
        q = engine.make_params(3, 60, 80, len(idx), call=6)
        gi = torch.from_numpy(idx.astype(np.int32)).cuda()
        engine.set_hyp_index(q, gi)
        engine.forward_device(sc, torch.from_numpy(ha[idx]).cuda(), q)
        np.testing.assert_array_equal(engine.read(api.BUF_SAMPLE_XY), xy_all[idx])
        np.testing.assert_array_equal(engine.read(api.BUF_HYPS), hyps_all[idx])
        fl = engine.read(api.BUF_EXACT_FLAGS).astype(bool)
        both_fast = ~fl & ~flags_all[idx]
        np.testing.assert_array_equal(engine.read(api.BUF_SCORES)[both_fast], scores_all[idx][both_fast])


def _same_record(batch_rec, single_rec, upto):
    """A frame of a batch is refined by one workgroup, a single call by a team: every discrete field identical, the
    refined pose equal to what the rounding of the LM sums -- their summation order differs -- becomes through the damped
    normal equations (measured <= 6e-10; the bar against the oracle is 1e-6)."""
    discrete = [api.RES_HYP, api.RES_EXPERT, api.RES_REF_STEPS, api.RES_INLIERS]
    np.testing.assert_array_equal(batch_rec[discrete], single_rec[discrete])
    # (the winner's exact score: the team sums the cells member by member, the selection kernel thread by thread)
    assert abs(batch_rec[api.RES_SCORE] - single_rec[api.RES_SCORE]) <= 1e-12 * max(1.0, abs(single_rec[api.RES_SCORE]))
    np.testing.assert_allclose(batch_rec[:upto], single_rec[:upto], rtol=0, atol=1e-8)


def test_batched_forward_equals_sequential_calls(engine):
    """esac_hip_forward_batch: frame b == the b-th of B consecutive single calls, BIT FOR BIT on the same refinement route: a
    batch of up to 32 frames refines every winner by a team of 8 like the single call does (round 5), a larger batch -- and
    any batch with teams switched off -- by one workgroup per frame; across the two routes the records agree to the rounding
    of the LM sums."""
    try:
        # (N * B <= 2048 throughout: beyond that the fp32 score stream runs with 4 wavefronts per hypothesis instead of 8 --
        # another summation order of the cells, so non-contender scores and the softmax statistics move in their last fp32 digit)
        # (exactly 8 members for the single calls: a batch's teams are 8 per frame, the single call's default is 10 on this grid --
        # another order of the LM sums)
        for B, team, N in ((12, api.REFINE_TEAM_EIGHT, 96), (12, 0, 96), (35, api.REFINE_TEAM_EIGHT, 48)):
            frames = [S.make_frame(70 + b, E=2, true_expert=b % 2) for b in range(B)]
            assigns = np.stack([S.gating_assignment(f, N, mode="gating") for f in frames])
            coords = torch.from_numpy(np.stack([f["coords"] for f in frames])).cuda()
            ha = torch.from_numpy(assigns).cuda()
            scores_b = torch.empty(B, N, dtype=torch.float64, device="cuda")
            p = engine.make_params(2, 60, 80, N, call=40)
            engine.set_refine_team(team)
            res_b = engine.forward_batch(coords, ha, p, scores_out=scores_b)
            batch_teams = team != 0 and B <= 32
            assert engine.refine_info()["mode"] == ("team" if batch_teams else "one_workgroup")
            for single_team in (0, api.REFINE_TEAM_EIGHT):
                engine.set_refine_team(single_team)
                for b in range(B if single_team == team else 4):
                    q = engine.make_params(2, 60, 80, N, call=40 + b)
                    s1 = torch.empty(N, dtype=torch.float64, device="cuda")
                    r1 = engine.forward_device(coords[b], ha[b], q, scores_out=s1)
                    assert engine.refine_info()["mode"] == ("team" if single_team else "one_workgroup")
                    if (single_team != 0) == batch_teams:  # the same route: the same bits
                        np.testing.assert_array_equal(res_b[b][:31], r1[:31], err_msg="B %d team %d single %d frame %d" % (B, team, single_team, b))
                        np.testing.assert_array_equal(scores_b[b].cpu().numpy(), s1.cpu().numpy())
                    else:  # the other summation order of the cells
                        _same_record(res_b[b], r1, 31)
                        np.testing.assert_allclose(scores_b[b].cpu().numpy(), s1.cpu().numpy(), rtol=1e-12, atol=0)
    finally:
        engine.set_refine_team(api.REFINE_TEAM_DEFAULT)
    # module-level API, shared maps for every frame
    import esac
    esac.set_seed(1305, 40)
    poses = torch.zeros(3, 4, 4)
    experts = esac.forward_batch(coords[0], ha[:3], poses, 0, 0, 525.0, 320.0, 240.0, 10.0, 100.0, 0.5, 100.0, 8)
    q = engine.make_params(2, 60, 80, N, call=41)
    r1 = engine.forward_device(coords[0], ha[1], q)
    np.testing.assert_allclose(poses[1].numpy().reshape(-1), r1[api.RES_POSE:api.RES_POSE + 16].astype(np.float32), rtol=0, atol=1e-6)
    assert experts[1] == int(r1[api.RES_EXPERT]) and esac.get_rng_state() == (1305, 43)


def test_debug_error_image_option(engine, oracle):
    """ESAC_BUF_WINNER_ERRS is only kept on request (esac_hip_set_debug); when kept it is the reprojection-error image
    of the refined pose (the last error pass of refineHyp, esac_util.h:445-452)."""
    f = S.make_frame(70)
    ha = S.gating_assignment(f, 64)
    sc, hat = torch.from_numpy(f["coords"]).cuda(), torch.from_numpy(ha).cuda()
    p = engine.make_params(1, 60, 80, 64, seed=3, call=1)
    engine.set_debug(keep_error_image=False)
    engine.forward_device(sc, hat, p)
    with pytest.raises(RuntimeError, match="esac_hip_set_debug"):
        engine.read(api.BUF_WINNER_ERRS)
    engine.set_debug(keep_error_image=True)
    try:
        res = engine.forward_device(sc, hat, p)
        errs = engine.read(api.BUF_WINNER_ERRS)
    finally:
        engine.set_debug(keep_error_image=False)
    pose = res[api.RES_RVEC:api.RES_RVEC + 6]
    pts = f["coords"][0].reshape(3, -1).T.copy()
    uv = oracle.project(pose[:3], pose[3:], f["focal"], f["focal"], f["ppx"], f["ppy"], pts).reshape(60, 80, 2)
    ys, xs = np.mgrid[0:60, 0:80]
    want = np.minimum(np.hypot(xs * 8 + 4 - uv[..., 0], ys * 8 + 4 - uv[..., 1]), 100.0)
    np.testing.assert_allclose(errs, want, rtol=0, atol=2e-2)  # fp32-accurate away from tau
    np.testing.assert_array_equal(errs < 10.0, want < 10.0 - 0)  # the inlier side is decided exactly


@pytest.mark.parametrize("max_tries", [0, 40, 16, 7])
def test_two_phase_sampling_matches_oracle(engine, oracle, max_tries):
    """More than 4096 hypotheses in flight take the throughput-shaped sampling: 16 tries of four hypotheses per
    wavefront first, the unaccepted rest one wavefront each from try 16 on.  Sampled cells, accepted try and poses
    must still be what the sequential loop of the reference produces -- incl. wrong-expert hypotheses that need
    hundreds of tries (phase 2) and exhausted budgets inside phase 1 (7, 16) and phase 2 (40)."""
    f = S.make_frame(120, E=4, true_expert=2)
    N = 4608
    ha = S.gating_assignment(f, N, mode="gating")
    sc, hat = torch.from_numpy(f["coords"]).cuda(), torch.from_numpy(ha).cuda()
    p = engine.make_params(4, 60, 80, N, seed=11, call=3, max_tries=max_tries)
    engine.sample(sc, hat, p)
    ref = oracle.forward(f["coords"], ha, seed=11, call=3, max_tries=max_tries)
    tries = engine.read(api.BUF_TRIES)
    np.testing.assert_array_equal(tries, ref["tries"])
    np.testing.assert_array_equal(engine.read(api.BUF_SAMPLE_XY), ref["sample_xy"])
    diff = np.abs(engine.read(api.BUF_HYPS) - ref["hyps"]).max(axis=1)
    # every ACCEPTED hypothesis agrees; a hypothesis whose budget ran out keeps the pose of its last, rejected try, and
    # on such a sample two P3P candidates can be equally bad for the 4th point: the two solvers may then keep different
    # ones (seen on 1 of 4608 at max_tries = 7) -- tolerated for at most one in a thousand of the exhausted ones
    assert (diff[tries >= 0] <= 1e-6).all()
    assert (diff[tries < 0] > 1e-6).sum() <= max(1, int((tries < 0).sum()) // 1000)
    if max_tries == 0:
        assert (tries >= 16).any() and (tries < 16).any()  # both phases delivered hypotheses
    else:
        assert (tries == -1).any()  # some budgets ran out


def test_large_batch_equals_sequential_calls(engine):
    """The same through forward_batch: 48 frames x 128 hypotheses (two-phase sampling, 4-wavefront score kernel) ==
    48 single calls (candidates of a try shared by lanes, 8-wavefront score kernel) in everything that is decided exactly: winner, its exact
    score, expert, refined pose, refinement trace.  (Selection probability and entropy are statistics of the fp32
    score stream, whose summation order follows the launch shape: equal to ~1e-6, not bit for bit.)"""
    B, N = 48, 128
    frames = [S.make_frame(130 + b, E=2, true_expert=b % 2) for b in range(B)]
    assigns = np.stack([S.gating_assignment(f, N, mode="gating") for f in frames])
    coords = torch.from_numpy(np.stack([f["coords"] for f in frames])).cuda()
    ha = torch.from_numpy(assigns).cuda()
    p = engine.make_params(2, 60, 80, N, call=500)
    res_b = engine.forward_batch(coords, ha, p)
    for b in range(0, B, 5):
        q = engine.make_params(2, 60, 80, N, call=500 + b)
        res_1 = engine.forward_device(coords[b], ha[b], q)
        _same_record(res_b[b], res_1, api.RES_PROB)
        assert res_b[b][api.RES_LM_ITERS] == res_1[api.RES_LM_ITERS]
        np.testing.assert_allclose(res_b[b][api.RES_PROB:api.RES_ENTROPY + 1], res_1[api.RES_PROB:api.RES_ENTROPY + 1], rtol=1e-5)


def test_more_experts_than_helper_class_bins(engine, oracle):
    """1500 experts (the per-expert counters that hand out the helpers of the screened search live in 1024 bins,
    esac_kernels.hip: expert_stats -- experts e and e + 1024 share one): scheduling only, every stage must still equal the
    oracle.  2048 hypotheses spread over all experts, a third of them on the true one; small try budget (a 12x16 grid of a
    wrong expert offers few consistent 4-point samples: the budget is what ends most of those searches)."""
    E = 1500
    f = S.make_frame(250, E=E, true_expert=1100, H=12, W=16, sub=40)
    rng = np.random.default_rng(5)
    ha = rng.integers(0, E, size=2048).astype(np.int64)
    ha[::3] = 1100
    res, ref = _both(engine, oracle, f["coords"], ha, focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=40, seed=1305, call=3, max_tries=700)
    _same(engine, res, ref)
    assert ref["expert"] == 1100 and (ref["tries"] < 0).sum() > 100  # budgets spent on wrong experts' maps


def _discrete(engine, res):
    return dict(tries=engine.read(api.BUF_TRIES), xy=engine.read(api.BUF_SAMPLE_XY), winner=int(res[api.RES_HYP]), steps=int(res[api.RES_REF_STEPS]),
                counts=engine.read(api.BUF_INLIER_COUNTS), imap=engine.read(api.BUF_INLIER_MAP), lm=int(res[api.RES_LM_ITERS]),
                rec=res.copy(), scores=engine.read(api.BUF_SCORES), flags=engine.read(api.BUF_EXACT_FLAGS))


@pytest.mark.parametrize("route", ["default", "fast", "one_workgroup", "stream", "tiled", "several_experts", "batch"])
@pytest.mark.parametrize("where", ["outlier_cell", "winner_inlier"])
def test_non_finite_scene_coordinates(engine, oracle, where, route):
    """A scene coordinate that is NaN / +Inf / -Inf -- what a diverged expert network emits.

    The REFERENCE: `std::min((float)cv::norm(curPt), maxReproj)` (esac_util.h:358) returns the NaN, so every score of that
    expert's hypotheses is NaN, softMax is NaN throughout and draw() keeps index 0 (esac_util.h:512-529): it refines hypothesis 0
    and returns that -- pinned against the reference's own sources in tests/test_oracle_vs_ref.py::
    test_non_finite_scene_coordinate_matches_reference.

    THIS implementation deliberately does not follow it there (deviation table): a cell that is not finite is a cell that
    can never be an inlier -- scored as an outlier at maxReproj, kept out of every inlier set (the refinement agrees with the
    reference on that: `NaN < tau` is false, esac_util.h:404) -- and the selection proceeds over finite scores.  The statement
    this test holds on every route: a non-finite cell behaves EXACTLY like the same cell at 1e30 (a coordinate the reference clamps
    to maxReproj, esac_util.h:358) -- same samples, same winner, same refinement trace, same pose -- and that call in turn agrees
    with the oracle on the 1e30 map as every other frame does."""
    E = 3 if route == "several_experts" else 1
    N = 400 if route == "several_experts" else 256
    f = S.make_frame(41, E=E, true_expert=E - 1)
    ha = S.gating_assignment(f, N, mode="gating")
    kw = dict(focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"], seed=1305, call=17)
    te = f["true_expert"]
    clean = oracle.forward(f["coords"], ha, **kw)
    if where == "winner_inlier":
        ys, xs = np.nonzero(clean["inlier_map"])
    else:
        ys, xs = np.nonzero(f["outlier_mask"] & (clean["inlier_map"] == 0))
    cells = [(int(ys[i]), int(xs[i])) for i in (0, len(ys) // 2, len(ys) - 1)]
    bad_maps = {}
    for name, vals in (("non_finite", (np.nan, np.inf, -np.inf)), ("huge", (1e30, 1e30, -1e30))):
        c = f["coords"].copy()
        for (y, x), v, ch in zip(cells, vals, (0, 1, 2)):
            c[te, ch, y, x] = v
        bad_maps[name] = c
    ref = oracle.forward(bad_maps["huge"], ha, **kw)  # (the oracle on the NaN map is the reference's "hypothesis 0" answer)
    mk = dict(kw)
    if route == "default":
        mk["exact_scores"] = "auto"
    elif route in ("stream", "tiled"):
        mk["score_shape"] = route
    if route == "one_workgroup":
        engine.set_refine_team(0)
    try:
        got = {}
        for name, c in bad_maps.items():
            sc, hat = torch.from_numpy(c).cuda(), torch.from_numpy(ha).cuda()
            p = engine.make_params(E, 60, 80, N, **mk)
            if route == "batch":
                other = torch.from_numpy(S.make_frame(42)["coords"]).cuda()
                recs = engine.forward_batch(torch.stack([other, sc, other]), torch.stack([hat, hat, hat]), engine.make_params(E, 60, 80, N, **dict(mk, call=16)))
                got[name] = dict(rec=recs[1].copy())
            else:
                got[name] = _discrete(engine, engine.forward_device(sc, hat, p))
    finally:
        engine.set_refine_team(api.REFINE_TEAM_DEFAULT)
    a, b = got["non_finite"], got["huge"]
    if route == "batch":
        np.testing.assert_array_equal(a["rec"][:31], b["rec"][:31])
        r_err, t_err = S.pose_errors(a["rec"][api.RES_POSE:api.RES_POSE + 16].reshape(4, 4), ref["pose"])
        assert int(a["rec"][api.RES_HYP]) == ref["winner"] and r_err <= 1e-4 and t_err <= 1e-3
        return
    for key in ("tries", "xy", "winner", "steps", "counts", "imap", "lm", "flags"):
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)
    assert np.isfinite(a["scores"]).all()
    np.testing.assert_allclose(a["scores"], b["scores"], rtol=0, atol=2e-3)
    np.testing.assert_allclose(a["rec"][:31], b["rec"][:31], rtol=0, atol=1e-9)
    for (y, x) in cells:
        assert a["imap"][y, x] == 0  # never an inlier
    # ... and the 1e30 call is an ordinary frame: held against the oracle
    np.testing.assert_array_equal(a["tries"], ref["tries"])
    np.testing.assert_array_equal(a["xy"], ref["sample_xy"])
    assert a["winner"] == ref["winner"] and a["steps"] == ref["ref_steps"]
    np.testing.assert_array_equal(a["counts"], ref["inlier_counts"])
    np.testing.assert_array_equal(a["imap"], ref["inlier_map"])
    r_err, t_err = S.pose_errors(a["rec"][api.RES_POSE:api.RES_POSE + 16].reshape(4, 4), ref["pose"])
    assert r_err <= 1e-4 and t_err <= 1e-3, (r_err, t_err)
