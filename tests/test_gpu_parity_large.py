"""GPU parity at the shapes of BASELINE.json configs[3] / configs[4] (and the N > 4096 launch shapes on a larger grid):
the FULL forward -- two-phase sampling, streaming score, select + exact re-score, refinement -- against the CPU oracle
on identical inputs and RNG key, stage by stage (tests/test_gpu_parity.py:_check_full).  The oracle needs seconds for
these on the GPU box's host cores."""
import numpy as np
import pytest
import torch

from esac_amd import api
from esac_amd import synthetic as S
from tests.test_gpu_parity import _check_full, _run_both

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", range(2))
def test_config4_12_experts_4096_hypotheses(engine, oracle, k):
    """configs[3]: 12 experts, gating active, 4096 hypotheses (single GPU here; the 4-GPU split is test_distributed_*)."""
    f = S.make_frame(200 + k, E=12, true_expert=(3 + 5 * k) % 12)
    ha = S.gating_assignment(f, 4096, mode="gating")
    res, ref = _run_both(engine, oracle, f, ha, call=k)
    _check_full(engine, res, ref)
    assert ref["tries"].max() > 64  # wrong-expert hypotheses went through the long sampling loop


def test_config5a_50_experts_16384_hypotheses(engine, oracle):
    """configs[4] at the native 60x80 grid: 50 experts, Dirichlet(0.3) gating, 16384 hypotheses -- k_sample_first x2,
    k_sample<64>, the many-hypotheses score kernel, select over 16384, refinement."""
    f = S.make_frame(201, E=50, true_expert=7)
    ha = S.gating_assignment(f, 16384, mode="dirichlet")
    ha[::97] = 7  # the Dirichlet draw may starve the true expert: keep ~170 hypotheses on it
    res, ref = _run_both(engine, oracle, f, ha, call=2)
    _check_full(engine, res, ref)
    assert ref["expert"] == 7


def test_config5b_shape_full_resolution_maps(engine, oracle):
    """The 5b stress shape at a size the oracle affords: 480x640 maps (subSampling = 1, esac.cpp:87-90 takes H, W from
    the tensor), 4 experts, 512 hypotheses: tiled scoring over 307,200 cells, refinement with the list in global memory."""
    f = S.make_frame(202, E=4, true_expert=1, H=480, W=640, sub=1)
    ha = S.gating_assignment(f, 512, mode="gating")
    res, ref = _run_both(engine, oracle, f, ha, call=3)
    r, t = _check_full(engine, res, ref)
    assert ref["ref_steps"] >= 1 and ref["inlier_counts"][0] > 100000


def test_many_hypotheses_on_a_larger_grid(engine, oracle):
    """N > 4096 on a 120x160 grid: two-phase sampling + the throughput-shaped score kernel + select + refinement (global
    correspondence list: 19,200 cells > the LDS capacity) in one call."""
    f = S.make_frame(203, E=3, true_expert=2, H=120, W=160, sub=4)
    ha = S.gating_assignment(f, 5000, mode="gating")
    res, ref = _run_both(engine, oracle, f, ha, call=4)
    _check_full(engine, res, ref)


def test_world_frame_coordinates_far_from_the_origin(engine, oracle):
    """Outdoor-style maps (Aachen / Dubrovnik live ~1e3 m from the world origin): the fp32 scoring stream works relative
    to each map's own origin, so ranking, band and winner behave as for a room at the origin."""
    f = S.make_frame(204, E=2, true_expert=1)
    off = np.array([1200.0, -800.0, 950.0], np.float32)
    f["coords"] = (f["coords"] + off[None, :, None, None]).astype(np.float32)
    ha = S.gating_assignment(f, 256, mode="gating")
    res, ref = _run_both(engine, oracle, f, ha, call=5)
    scores = engine.read(api.BUF_SCORES)
    flags = engine.read(api.BUF_EXACT_FLAGS).astype(bool)
    # float32 coordinates at 1e3 m carry 6e-5 m of quantisation: the reference itself sees them; the fp32 stream must
    # stay within the band's head-room of the reference arithmetic
    assert np.abs(scores[~flags] - ref["scores"][~flags]).max() < 5e-3
    np.testing.assert_allclose(scores[flags], ref["scores"][flags], rtol=0, atol=1e-7)
    assert int(res[api.RES_HYP]) == ref["winner"] and flags[ref["winner"]]
    assert int(res[api.RES_REF_STEPS]) == ref["ref_steps"]
    np.testing.assert_array_equal(engine.read(api.BUF_INLIER_COUNTS), ref["inlier_counts"])
    r, t = S.pose_errors(res[api.RES_POSE:api.RES_POSE + 16].reshape(4, 4), ref["pose"])
    assert r <= 1e-4 and t <= 1e-3, (r, t)


@pytest.mark.parametrize("shape", [(60, 80, 8, 5, 700), (45, 64, 10, 1, 300), (128, 160, 4, 70, 2500)])
def test_tiled_score_equals_the_per_hypothesis_stream(engine, oracle, shape):
    """The two shapes of the fp32 ranking score (tile-stationary with hypotheses bucketed by expert vs one hypothesis per
    workgroup) on grids whose last sub-tile is partly filled (4800, 2880 cells) or exactly filled (20480), with empty
    experts and experts of several chunks: same scores to fp32 rounding, both within the band of the reference arithmetic."""
    H, W, sub, E, N = shape
    f = S.make_frame(210, E=E, true_expert=E // 2, H=H, W=W, sub=sub)
    ha = S.gating_assignment(f, N, mode="dirichlet" if E > 1 else "single")
    ha[::3] = E // 2
    sc, hat = torch.from_numpy(f["coords"]).cuda(), torch.from_numpy(ha).cuda()
    kw = dict(shift_x=f["shift"][0], shift_y=f["shift"][1], focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=sub, seed=3, call=9)
    ref = oracle.forward(f["coords"], ha, **kw)
    out = {}
    for mode in ("stream", "tiled"):
        p = engine.make_params(E, H, W, N, score_shape=mode, **kw)
        engine.sample(sc, hat, p)
        engine.score(sc, hat, p)
        engine.select(sc, hat, p)
        out[mode] = (engine.read(api.BUF_SCORES), engine.read(api.BUF_EXACT_FLAGS).astype(bool))
        res = engine.forward_device(sc, hat, p)
        assert int(res[api.RES_HYP]) == ref["winner"], mode
    (s_s, f_s), (s_t, f_t) = out["stream"], out["tiled"]
    both = ~f_s & ~f_t
    assert both.sum() > N // 2
    assert np.abs(s_s[both] - s_t[both]).max() <= 2e-4
    d = np.abs(s_t[~f_t] - ref["scores"][~f_t])
    assert np.sort(d)[-3:].max() <= 2 * 100.0 / (H * W) + 2e-3 and np.median(d) <= 1e-4
    np.testing.assert_allclose(s_t[f_t], ref["scores"][f_t], rtol=0, atol=1e-7)


def test_more_pending_hypotheses_than_the_screened_chain_used_to_launch_wavefronts(engine, oracle):
    """140,000 hypotheses over two experts with the screened chain starting at try 0: more pending hypotheses than the
    131072 wavefronts the chain's launch was once capped at -- wavefront L serves list entry L % count, so every entry
    needs a wavefront of its own.  Sampled cells and accepted tries of ALL hypotheses against the oracle (a small try
    budget keeps the oracle's wrong-expert searches short; an exhausted budget is -1 on both sides)."""
    N = 140000
    f = S.make_frame(230, E=2, true_expert=1)
    ha = S.gating_assignment(f, N, mode="gating")
    sc, hat = torch.from_numpy(f["coords"]).cuda(), torch.from_numpy(ha).cuda()
    kw = dict(focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"], seed=9, call=2, max_tries=48)
    engine.sample(sc, hat, engine.make_params(2, 60, 80, N, **kw))
    tries, xy = engine.read(api.BUF_TRIES), engine.read(api.BUF_SAMPLE_XY)
    ref = oracle.sample(f["coords"], ha, **kw) if hasattr(oracle, "sample") else oracle.forward(f["coords"], ha, max_ref_steps=0, **kw)
    np.testing.assert_array_equal(tries, ref["tries"])
    np.testing.assert_array_equal(xy, ref["sample_xy"])
    assert (tries == -1).any() and (tries >= 32).any()


def test_large_beta_tau_on_a_tiled_size_grid(engine, oracle):
    """beta * tau beyond the range of the tile kernel's folded sigmoid constant (2^(-k tau), k = |beta| log2 e, underflows
    from k tau ~ 126): the C ABI must route such calls to the per-hypothesis stream -- also when the tiled shape is ASKED
    for -- and the scores must stay finite and within the band of the reference arithmetic."""
    H, W, sub, E, N = 128, 160, 4, 2, 300
    f = S.make_frame(212, E=E, true_expert=1, H=H, W=W, sub=sub)
    ha = S.gating_assignment(f, N, mode="gating")
    sc, hat = torch.from_numpy(f["coords"]).cuda(), torch.from_numpy(ha).cuda()
    for beta, tau in ((9.0, 10.0), (0.5, 400.0), (-9.0, 10.0)):
        kw = dict(shift_x=f["shift"][0], shift_y=f["shift"][1], focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=sub, seed=3, call=9,
                  inlier_beta=beta, inlier_thresh=tau, max_reproj=max(100.0, 2 * tau))
        ref = oracle.forward(f["coords"], ha, **kw)
        for mode in ("auto", "tiled"):
            res = engine.forward_device(sc, hat, engine.make_params(E, H, W, N, score_shape=mode, **kw))
            scores, flags = engine.read(api.BUF_SCORES), engine.read(api.BUF_EXACT_FLAGS).astype(bool)
            assert np.isfinite(scores).all(), (beta, tau, mode)
            d = np.abs(scores[~flags] - ref["scores"][~flags])
            assert np.sort(d)[-3:].max() <= 2 * 100.0 / (H * W) + 2e-3, (beta, tau, mode, d.max())
            assert int(res[api.RES_HYP]) == ref["winner"], (beta, tau, mode)


def test_packed_map_copy_for_the_sampler(engine, oracle):
    """ESAC_FLAG_PACK_MAPS (taken by default for maps far beyond the caches): the sampler gathers (x,y,z) records from a
    packed copy of the maps -- sampled cells, accepted tries and everything downstream must not change.  5000 hypotheses
    over 3 experts: first-phase passes, prescreen / decide / commit and the resume kernel all read through the copy."""
    f = S.make_frame(220, E=3, true_expert=1, H=120, W=160, sub=4)
    ha = S.gating_assignment(f, 5000, mode="dirichlet")
    ha[::5] = 1
    sc, hat = torch.from_numpy(f["coords"]).cuda(), torch.from_numpy(ha).cuda()
    kw = dict(shift_x=f["shift"][0], shift_y=f["shift"][1], focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=4, seed=5, call=1)
    ref = oracle.forward(f["coords"], ha, **kw)
    res = engine.forward_device(sc, hat, engine.make_params(3, 120, 160, 5000, pack_maps=True, **kw))
    _check_full(engine, res, ref)
    # and on a small single-frame call (the latency-shaped sampler)
    ha2 = ha[:300].copy()
    ref2 = oracle.forward(f["coords"], ha2, **kw)
    res2 = engine.forward_device(sc, torch.from_numpy(ha2).cuda(), engine.make_params(3, 120, 160, 300, pack_maps=True, **kw))
    _check_full(engine, res2, ref2)


def test_config5b_true_50_experts_16384_hypotheses_full_resolution(engine, oracle):
    """BASELINE configs[4] itself -- 50 experts, Dirichlet(0.3) gating, 16384 hypotheses, 480x640 maps (subSampling 1) --
    the one call in which the packed map copy of the sampler (default trigger: E*P*12 >= 32 MB), the tile-stationary score
    with dozens of chunks per expert, the 16-way split selection and the cooperating refinement workgroups all run
    together: the FULL forward against the oracle, stage by stage (5.0e9 cell evaluations: ~20 s of the host's cores)."""
    E, N, H, W = 50, 16384, 480, 640
    f = S.make_frame(230, E=E, true_expert=11, H=H, W=W, sub=1)
    ha = S.gating_assignment(f, N, mode="dirichlet")
    ha[::131] = 11  # the Dirichlet draw may starve the true expert: keep ~125 hypotheses on it
    assert E * H * W * 12 >= 32 << 20  # the default want_pack trigger (esac_capi.hip) fires: no flag is passed below
    res, ref = _run_both(engine, oracle, f, ha, call=6)
    _check_full(engine, res, ref)
    assert ref["expert"] == 11 and ref["tries"].max() > 64
    assert ref["ref_steps"] >= 1 and ref["inlier_counts"][0] > 100000  # 38 cooperating workgroups refine 307,200 cells
    engine.check()  # neither an out-of-range assignment nor a barrier time-out


def test_more_stragglers_than_resident_wavefronts(engine, oracle):
    """3000 hypotheses, nearly all on WRONG experts (each needs ~10^3 tries): more pending hypotheses than the 2304
    wavefronts of the migrating screened search (esac_kernels.hip: k_sample_prescreen<true>) -- every list entry must be
    started by the wavefront that owns it or by one that found it, none may be left to the commit kernel unscreened."""
    f = S.make_frame(240, E=4, true_expert=1)
    ha = np.array([0, 2, 3], np.int64)[np.arange(3000) % 3]
    ha[::50] = 1
    res, ref = _run_both(engine, oracle, f, ha, call=7)
    _check_full(engine, res, ref)
    assert (ref["tries"] > 32).sum() > 2500 and ref["expert"] == 1
