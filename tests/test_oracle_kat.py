"""Known-answer tests that pin the CPU oracle (the reference ships no tests or golden vectors).

Each block checks one OpenCV stand-in or one reference-level rule against an answer that is known
analytically or from an independent computation (numpy), so that a slip in the from-memory
restatement shows up here rather than as a silent parity claim.
"""
import math

import numpy as np
import pytest

from esac_amd import synthetic as S

FX = FY = 525.0
CX, CY = 320.0, 240.0


def _rand_pose(rng, max_angle=1.0):
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    r = axis * rng.uniform(0.05, max_angle)
    t = rng.uniform(-0.5, 0.5, size=3) + np.array([0, 0, 4.0])
    return r, t


def _project(oracle, r, t, pts):
    R = oracle.rodrigues_vec2mat(r)
    Xc = (R @ pts.T).T + t
    return np.stack([FX * Xc[:, 0] / Xc[:, 2] + CX, FY * Xc[:, 1] / Xc[:, 2] + CY], 1)


# ---------------------------------------------------------------- RNG
def test_philox_known_answers(oracle):
    """Random123 kat_vectors for philox4x32-10."""
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, want in kat:
        assert [int(v) for v in oracle.philox(ctr, key)] == want


def test_sampling_range_and_distinctness(oracle):
    """irand(0, imW-1) has an exclusive bound (thread_rand.cpp:68-71): last row/column never drawn;
    the four cells of a try are distinct (esac_util.h:170-176)."""
    W, H = 80, 60
    seen_x, seen_y = set(), set()
    for h in range(40):
        for t in range(40):
            xy = oracle.draw_cells(1305, 3, h, t, W, H)
            assert len({(int(x), int(y)) for x, y in xy}) == 4
            seen_x |= {int(v) for v in xy[:, 0]}
            seen_y |= {int(v) for v in xy[:, 1]}
    assert max(seen_x) == W - 2 and min(seen_x) == 0
    assert max(seen_y) == H - 2 and min(seen_y) == 0
    # tiny grid: duplicates are frequent, the redraw rule must still deliver 4 distinct cells
    for t in range(200):
        xy = oracle.draw_cells(7, 0, 0, t, 4, 4)  # cells in [0,2]x[0,2]
        assert len({(int(x), int(y)) for x, y in xy}) == 4 and xy.max() <= 2


# ---------------------------------------------------------------- polynomial + P3P
def test_quartic_against_numpy_roots(oracle):
    rng = np.random.default_rng(0)
    checked = 0
    for _ in range(300):
        roots = np.sort(rng.uniform(-3, 3, size=4))
        if np.min(np.diff(roots)) < 0.2:
            continue
        c = np.poly(roots) * rng.uniform(0.5, 2.0)
        got = np.sort(oracle.solve_deg4(*c))
        assert len(got) == 4
        np.testing.assert_allclose(got, roots, atol=1e-7)
        checked += 1
    assert checked > 50
    # two real + two complex roots
    c = np.poly([1.0, -2.0, 0.5 + 1.0j, 0.5 - 1.0j]).real
    np.testing.assert_allclose(np.sort(oracle.solve_deg4(*c)), [-2.0, 1.0], atol=1e-9)
    # no real root
    assert len(oracle.solve_deg4(1.0, 0.0, 3.0, 0.0, 5.0)) == 0


def test_p3p_recovers_ground_truth(oracle):
    """Exact synthetic correspondences: the GT pose is among the 3-point solutions to 1e-9 and the
    4-point variant selects it (smallest reprojection error of point 3)."""
    rng = np.random.default_rng(1)
    d3, d4 = [], []
    for _ in range(400):
        r, t = _rand_pose(rng)
        pts = rng.uniform(-1.5, 1.5, size=(4, 3))
        img = _project(oracle, r, t, pts)
        Rs, ts = oracle.p3p_all(pts[:3], img[:3], FX, FY, CX, CY)
        R_gt = oracle.rodrigues_vec2mat(r)
        assert len(Rs) >= 1
        d3.append(min(max(np.abs(R - R_gt).max(), np.abs(tt - t).max()) for R, tt in zip(Rs, ts)))
        for R in Rs:  # every solution is a proper rotation
            np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-9)
            assert np.linalg.det(R) > 0.999
        ok, rv, tv = oracle.p3p(pts, img, FX, FY, CX, CY)
        assert ok
        d4.append(max(np.abs(rv - r).max(), np.abs(tv - t).max()))
    # the closed-form quartic (Ferrari) loses digits near multiple roots -- the published
    # algorithm's known weakness -- hence a distribution, not a single bound
    for d in (np.array(d3), np.array(d4)):
        assert np.median(d) < 1e-8 and np.percentile(d, 90) < 1e-5 and d.max() < 5e-2, np.percentile(d, [50, 90, 100])


def test_p3p_solutions_satisfy_constraints_on_noisy_data(oracle):
    rng = np.random.default_rng(2)
    worst = []
    for _ in range(200):
        r, t = _rand_pose(rng)
        pts = rng.uniform(-1.5, 1.5, size=(3, 3))
        img = _project(oracle, r, t, pts) + rng.normal(0, 2.0, size=(3, 2))
        Rs, ts = oracle.p3p_all(pts, img, FX, FY, CX, CY)
        for R, tt in zip(Rs, ts):
            Xc = (R @ pts.T).T + tt
            assert (Xc[:, 2] > 0).all()  # positive lengths only (x, y > 0 in Gao's parametrisation)
            uv = np.stack([FX * Xc[:, 0] / Xc[:, 2] + CX, FY * Xc[:, 1] / Xc[:, 2] + CY], 1)
            worst.append(np.abs(uv - img).max())
    worst = np.array(worst)
    assert len(worst) > 200  # usually 2 solutions per sample
    assert np.median(worst) < 1e-7 and np.percentile(worst, 90) < 1e-3 and worst.max() < 5.0, np.percentile(worst, [50, 90, 100])


# ---------------------------------------------------------------- Rodrigues / projection / pose2trans
def test_rodrigues_round_trip_and_jacobian(oracle):
    rng = np.random.default_rng(3)
    for _ in range(100):
        r, _ = _rand_pose(rng, max_angle=3.0)
        R, J = oracle.rodrigues_vec2mat(r, jac=True)
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-12)
        np.testing.assert_allclose(oracle.rodrigues_mat2vec(R), r, atol=1e-9)
        # closed form vs scipy-free reference: R = exp([r]x)
        th = np.linalg.norm(r)
        k = r / th
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        np.testing.assert_allclose(R, np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K, atol=1e-12)
        for i in range(3):  # J[i] = d vec(R) / d r_i, central differences
            d = np.zeros(3)
            d[i] = 1e-6
            num = (oracle.rodrigues_vec2mat(r + d) - oracle.rodrigues_vec2mat(r - d)).reshape(-1) / 2e-6
            np.testing.assert_allclose(J[i], num, atol=1e-8)
    R0, J0 = oracle.rodrigues_vec2mat(np.zeros(3), jac=True)
    np.testing.assert_array_equal(R0, np.eye(3))
    assert J0[0, 5] == -1 and J0[0, 7] == 1 and J0[1, 2] == 1 and J0[1, 6] == -1 and J0[2, 1] == -1 and J0[2, 3] == 1


def test_projection_semantics(oracle):
    """fp64 compute, float output, `z ? 1/z : 1`, NO cheirality test (cvProjectPoints2)."""
    pts = np.array([[0.1, -0.2, 2.0], [0.3, 0.1, -2.0], [0.5, 0.25, 0.0]], np.float32)
    uv = oracle.project(np.zeros(3), np.zeros(3), FX, FY, CX, CY, pts)
    assert uv.dtype == np.float32
    np.testing.assert_allclose(uv[0], [FX * 0.05 + CX, FY * -0.1 + CY], rtol=1e-6)
    np.testing.assert_allclose(uv[1], [FX * -0.15 + CX, FY * -0.05 + CY], rtol=1e-6)  # behind the camera: still projected
    np.testing.assert_allclose(uv[2], [FX * 0.5 + CX, FY * 0.25 + CY], rtol=1e-6)      # z == 0 -> scale 1


def test_pose2trans_is_the_rigid_inverse(oracle):
    rng = np.random.default_rng(4)
    for _ in range(50):
        r, t = _rand_pose(rng, 3.0)
        T = oracle.pose2trans(np.concatenate([r, t]))
        R = oracle.rodrigues_vec2mat(r)
        M = np.eye(4)
        M[:3, :3], M[:3, 3] = R, t
        np.testing.assert_allclose(T @ M, np.eye(4), atol=1e-12)
        np.testing.assert_allclose(T[:3, :3], R.T, atol=1e-12)
        np.testing.assert_allclose(T[:3, 3], -R.T @ t, atol=1e-12)


# ---------------------------------------------------------------- LM refit
def test_lm_recovers_ground_truth_on_noise_free_data(oracle):
    rng = np.random.default_rng(5)
    for _ in range(30):
        r, t = _rand_pose(rng)
        pts = rng.uniform(-1.5, 1.5, size=(200, 3)).astype(np.float32)
        img = _project(oracle, r, t, pts.astype(np.float64)).astype(np.float32)
        start = np.concatenate([r + rng.normal(0, 0.02, 3), t + rng.normal(0, 0.05, 3)])
        pose, iters = oracle.lm_pnp(pts, img, FX, FY, CX, CY, start)
        assert 1 <= iters <= 20
        np.testing.assert_allclose(pose[:3], r, atol=2e-5)  # image points were rounded to float32
        np.testing.assert_allclose(pose[3:], t, atol=2e-4)


def test_lm_minimises_reprojection_error_on_noisy_data(oracle):
    rng = np.random.default_rng(6)
    r, t = _rand_pose(rng)
    pts = rng.uniform(-1.5, 1.5, size=(300, 3)).astype(np.float32)
    img = (_project(oracle, r, t, pts.astype(np.float64)) + rng.normal(0, 1.0, (300, 2))).astype(np.float32)
    start = np.concatenate([r + 0.02, t - 0.05])
    pose, _ = oracle.lm_pnp(pts, img, FX, FY, CX, CY, start)

    def cost(q):
        return float(((_project(oracle, q[:3], q[3:], pts.astype(np.float64)) - img) ** 2).sum())
    c0 = cost(pose)
    assert c0 < cost(start)
    for k in range(6):  # local minimum: no coordinate direction improves the cost
        for s in (-1e-5, 1e-5):
            q = pose.copy()
            q[k] += s
            assert cost(q) >= c0 - 1e-7


# ---------------------------------------------------------------- reference-level rules
def _perfect_frame(oracle, noise=0.0, outliers=0.0, k=0):
    return S.make_frame(k, noise=noise, outlier_frac=outliers)


def test_score_known_answers(oracle):
    """Perfect map + true pose: every error ~0 -> score = alpha*(1 - sigmoid(-beta*tau)) = 99.3307...;
    all cells at maxReproj: alpha*(1 - sigmoid(beta*(100-10))) ~ 2.9e-18."""
    f = _perfect_frame(oracle)
    ha = S.gating_assignment(f, 16)
    o = oracle.forward(f["coords"], ha)
    want = 100.0 * (1.0 - 1.0 / (1.0 + math.exp(5.0)))
    assert abs(o["scores"].max() - want) < 0.05  # float32 coordinates leave ~1e-3 px errors
    # upper bound with the reference's float scale factor alpha/cols/rows (esac_util.h:256)
    scale = np.float32(np.float32(np.float32(100.0) / np.float32(80)) / np.float32(60))
    assert o["scores"].max() <= 4800 * float(scale) * (1.0 - 1.0 / (1.0 + math.exp(5.0))) + 1e-9
    r_err, t_err = S.pose_errors(o["pose"], f["gt_pose"])
    assert r_err < 1e-5 and t_err < 1e-4
    # garbage map: a single expert predicting coordinates unrelated to the image
    g = S.make_frame(1, E=2, true_expert=1)
    coords = g["coords"][:1].copy()  # expert 0 = garbage
    o = oracle.forward(coords, np.zeros(8, np.int64), max_tries=2000)
    assert o["scores"].max() < 5.0


def test_forward_recovers_pose_on_default_synthetic_frames(oracle):
    errs = []
    for k in range(8):
        f = S.make_frame(k)
        ha = S.gating_assignment(f, 256)
        o = oracle.forward(f["coords"], ha, call=k)
        errs.append(S.pose_errors(o["pose"], f["gt_pose"]))
        assert o["expert"] == 0 and 0 <= o["winner"] < 256
        assert o["ref_steps"] >= 1
        c = o["inlier_counts"]
        c = c[c >= 0]
        assert (np.diff(c[:-1]) > 0).all()  # refinement continues only while the count grows (esac_util.h:417)
        assert c[-1] <= c[:-1].max()
        assert o["inlier_map"].sum() == c[-2]  # map of the last ACCEPTED set
        assert abs(o["probs"].sum() - 1.0) < 1e-12
        assert o["winner"] == int(np.argmax(o["scores"]))
    r = np.array([e[0] for e in errs])
    t = np.array([e[1] for e in errs])
    assert np.median(r) < math.radians(0.5) and np.median(t) < 0.02


def test_budget_exhaustion_and_constant_map(oracle):
    """A constant map makes every P3P fail: the hypothesis stays the zero pose (esac_util.h:107-111) and the
    try budget runs out (tries = -1); the call still returns a pose."""
    coords = np.ones((1, 3, 12, 16), np.float32)
    o = oracle.forward(coords, np.zeros(4, np.int64), max_tries=50)
    assert (o["tries"] == -1).all()
    np.testing.assert_array_equal(o["hyps"], 0.0)
    assert np.isfinite(o["pose"]).all()


def test_strided_inputs(oracle):
    """accessor<> honours strides, incl. the stride-0 expand() of test_esac.py:171-173."""
    f = S.make_frame(2, E=3, true_expert=2)
    ha = np.broadcast_to(np.array([2], np.int64), (32,))
    assert ha.strides == (0,)
    a = oracle.forward(f["coords"], ha)
    b = oracle.forward(f["coords"], np.full(32, 2, np.int64))
    np.testing.assert_array_equal(a["pose"], b["pose"])
    big = np.zeros((3, 3, 60, 160), np.float32)
    big[..., ::2] = f["coords"]
    c = oracle.forward(big[..., ::2], np.full(32, 2, np.int64))
    np.testing.assert_array_equal(a["pose"], c["pose"])


def test_shift_and_subsampling(oracle):
    """createSampling: px = x*sub + sub/2 - shift (integer arithmetic, esac_util.h:64-66)."""
    f = S.make_frame(3, shift=(3, -2))
    ha = S.gating_assignment(f, 64)
    o = oracle.forward(f["coords"], ha, shift_x=3, shift_y=-2)
    r_err, t_err = S.pose_errors(o["pose"], f["gt_pose"])
    assert r_err < math.radians(1.0) and t_err < 0.03
    wrong = oracle.forward(f["coords"], ha, shift_x=0, shift_y=0)
    assert S.pose_errors(wrong["pose"], f["gt_pose"])[1] > t_err


def test_rng_key_independence(oracle):
    f = S.make_frame(4)
    ha = S.gating_assignment(f, 64)
    a = oracle.forward(f["coords"], ha, seed=1305, call=0)
    b = oracle.forward(f["coords"], ha, seed=1305, call=0, num_threads=1)
    c = oracle.forward(f["coords"], ha, seed=1305, call=1)
    np.testing.assert_array_equal(a["sample_xy"], b["sample_xy"])  # independent of the thread count
    np.testing.assert_array_equal(a["pose"], b["pose"])
    assert (a["sample_xy"] != c["sample_xy"]).any()
    # shard independence: the second half evaluated on its own with global indices
    idx = np.arange(32, 64, dtype=np.int32)
    d = oracle.forward(f["coords"], ha[32:], seed=1305, call=0, hyp_index=idx)
    np.testing.assert_array_equal(d["sample_xy"], a["sample_xy"][32:])
    np.testing.assert_array_equal(d["scores"], a["scores"][32:])


def test_argument_errors(oracle):
    with pytest.raises(RuntimeError):
        oracle.forward(np.zeros((1, 3, 2, 2), np.float32), np.zeros(4, np.int64))
    with pytest.raises(RuntimeError):
        oracle.forward(np.zeros((1, 3, 8, 8), np.float32), np.full(4, 3, np.int64))


# ---------------------------------------------------------------- training path of the oracle
def test_backward_score_path_matches_finite_differences(oracle):
    """Oracle esac_backward with max_ref_steps = 0 (no re-fit: path I vanishes, refined pose = initial pose): moving a
    coordinate of a cell nobody sampled changes the expected loss only through the scores, an exact derivative in the
    reference's formulas (softmax x sigmoid x d error / d point) -- central differences of the returned loss agree."""
    from esac_amd import synthetic as S
    f = S.make_frame(300, H=24, W=32, sub=20)
    ha = S.gating_assignment(f, 24)
    gt = f["gt_pose"].astype(np.float32)
    gt[:3, 3] += np.float32(0.05)
    kw = dict(focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"], inlier_alpha=3.0, seed=5, call=2, max_ref_steps=0)
    g = np.zeros_like(f["coords"])
    out = oracle.backward(f["coords"], g, ha, gt, **kw)
    assert (out["probs"] >= 1e-3).sum() >= 5
    sampled = set(map(tuple, out["sample_xy"].reshape(-1, 2)))
    checked = 0
    for idx in np.argsort(-np.abs(g[0]).reshape(-1)):
        c, rem = divmod(int(idx), 24 * 32)
        y, x = divmod(rem, 32)
        if (x, y) in sampled:
            continue
        h = 1e-3
        vals = []
        for s in (+1, -1):
            pert = f["coords"].copy()
            pert[0, c, y, x] += s * h
            vals.append(oracle.backward(pert, np.zeros_like(g), ha, gt, **kw)["loss"])
        fd = (vals[0] - vals[1]) / (2 * h)
        assert abs(fd - g[0, c, y, x]) <= 0.02 * abs(g[0, c, y, x]) + 1e-6, (fd, g[0, c, y, x])
        checked += 1
        if checked == 4:
            break
    assert checked == 4


def test_backward_accumulates_and_is_deterministic(oracle):
    from esac_amd import synthetic as S
    f = S.make_frame(301, H=24, W=32, sub=20)
    ha = S.gating_assignment(f, 16)
    gt = f["gt_pose"].astype(np.float32)
    kw = dict(focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"], seed=5, call=3)
    g1 = np.zeros_like(f["coords"])
    a = oracle.backward(f["coords"], g1, ha, gt, **kw)
    g2 = g1.copy()
    b = oracle.backward(f["coords"], g2, ha, gt, **kw)
    assert a["loss"] == b["loss"] and np.abs(g1).max() > 0
    np.testing.assert_allclose(g2, 2 * g1, rtol=1e-5, atol=1e-9)  # float `+=` into the caller's tensor (esac.cpp:491-508)
    g3 = np.zeros_like(g1)
    c = oracle.backward(f["coords"], g3, ha, gt, num_threads=1, **kw)
    assert c["loss"] == a["loss"]
    np.testing.assert_array_equal(g3, g1)  # thread-count independent
