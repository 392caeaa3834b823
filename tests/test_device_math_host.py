"""The kernels' own math headers (esac_amd/csrc/pose_math.hpp, lm_math.hpp), compiled for the HOST by
tests/native/build.py, checked against the oracle on the CPU.  This is the same source the GPU runs
(different back end), so formula slips are caught without a GPU; the -m gpu tests then check the device build.
"""
import ctypes as C

import numpy as np
import pytest

from esac_amd import synthetic as S

FX = FY = 525.0
CX, CY = 320.0, 240.0


@pytest.fixture(scope="module")
def probe():
    from tests.native import build
    lib = C.CDLL(build.build())
    d, vp = C.c_double, C.c_void_p
    lib.probe_p3p.argtypes = [vp, vp, d, d, d, d, vp, vp, vp]
    lib.probe_quartic.argtypes = [d, d, d, d, d, vp]
    lib.probe_rodrigues.argtypes = [vp, vp, vp]
    lib.probe_mat2vec.argtypes = [vp, vp]
    lib.probe_exact_err.argtypes = [vp, vp, d, d, d, d, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]
    lib.probe_exact_err.restype = C.c_float
    lib.probe_lm_solve6.argtypes = [vp, vp, d, vp]
    lib.probe_lm_normal.argtypes = [vp, vp, C.c_int, vp, d, d, d, d, vp, vp, vp]
    lib.probe_lm_normal_moments.argtypes = [vp, vp, C.c_int, vp, d, d, d, vp, vp, vp]
    f = C.c_float
    lib.probe_pose_loss.argtypes = [vp, vp, d, d, d]
    lib.probe_pose_loss.restype = d
    lib.probe_pose_dloss.argtypes = [vp, vp, d, d, d, vp]
    lib.probe_dproject_dobj.argtypes = [f, f, f, f, f, vp, vp, f, f, f, f, vp]
    lib.probe_norm_jac_row.argtypes = [vp, vp, f, f, f, f, f, f, f, f, f, vp]
    lib.probe_inv_spd6.argtypes = [vp, vp]
    lib.probe_pinv_sym6.argtypes = [vp, vp]
    lib.probe_chain_diff.argtypes = [vp]
    lib.probe_chain_diff.restype = d
    lib.probe_pose_chain.argtypes = [vp, vp, vp, vp]
    lib.probe_transform_t_diff.argtypes = [vp, vp]
    lib.probe_transform_t_diff.restype = d
    lib.probe_lane_step.argtypes = [vp, vp, d, vp, vp, vp]
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _probe_p3p(lib, obj, img):
    r, t, R = np.zeros(3), np.zeros(3), np.zeros(9)
    ok = lib.probe_p3p(_p(obj), _p(img), FX, FY, CX, CY, _p(r), _p(t), _p(R))
    return bool(ok), r, t


def test_p3p_matches_oracle_on_sampled_minimal_sets(oracle, probe):
    """Same minimal sets the forward path draws (good expert AND garbage experts): the GPU formulation
    (Gao + Ferrari + triad/Newton alignment) must land on the oracle's pose (Gao + Ferrari + Horn/Jacobi)."""
    f = S.make_frame(100, E=10, true_expert=0)
    ha = S.gating_assignment(f, 512, mode="gating")
    ref = oracle.forward(f["coords"], ha)
    worst = 0.0
    for h in range(512):
        e = ha[h]
        xy = ref["sample_xy"][h]
        obj = np.array([[f["coords"][e, c, y, x] for c in range(3)] for x, y in xy], np.float64)
        img = np.array([[x * 8 + 4, y * 8 + 4] for x, y in xy], np.float64)
        ok, r, t = _probe_p3p(probe, obj, img)
        assert ok == (ref["tries"][h] >= 0 or np.any(ref["hyps"][h] != 0))
        worst = max(worst, np.abs(np.concatenate([r, t]) - ref["hyps"][h]).max())
    assert worst < 1e-7, worst


def test_p3p_random_problems(oracle, probe):
    rng = np.random.default_rng(11)
    worst = 0.0
    n_ok = 0
    for _ in range(500):
        obj = rng.uniform(-2, 2, size=(4, 3))
        obj[:, 2] += 4.0
        img = rng.uniform([0, 0], [640, 480], size=(4, 2))  # arbitrary image points: many have no / several solutions
        ok_o, r_o, t_o = oracle.p3p(obj, img, FX, FY, CX, CY)
        ok_p, r_p, t_p = _probe_p3p(probe, obj, img)
        assert ok_o == ok_p
        if ok_o:
            n_ok += 1
            worst = max(worst, np.abs(np.concatenate([r_p - r_o, t_p - t_o])).max())
    assert n_ok > 50
    assert worst < 1e-6, worst


def test_quartic_matches_oracle_bitwise(oracle, probe):
    rng = np.random.default_rng(12)
    for _ in range(300):
        c = rng.normal(size=5)
        got = np.zeros(4)
        n = probe.probe_quartic(*[float(v) for v in c], _p(got))
        want = oracle.solve_deg4(*c)
        assert n == len(want)
        np.testing.assert_allclose(got[:n], want, rtol=1e-12, atol=1e-12)


def test_rodrigues_matches_oracle(oracle, probe):
    rng = np.random.default_rng(13)
    for _ in range(100):
        r = rng.normal(size=3) * rng.uniform(0.01, 1.5)
        R, J = np.zeros(9), np.zeros(27)
        probe.probe_rodrigues(_p(r), _p(R), _p(J))
        Ro, Jo = oracle.rodrigues_vec2mat(r, jac=True)
        np.testing.assert_allclose(R.reshape(3, 3), Ro, atol=1e-15)
        np.testing.assert_allclose(J.reshape(3, 9), Jo, atol=1e-14)
        back = np.zeros(3)
        probe.probe_mat2vec(_p(R), _p(back))
        np.testing.assert_allclose(back, oracle.rodrigues_mat2vec(Ro), atol=1e-15)


def test_exact_error_is_bit_identical_to_oracle(oracle, probe):
    """project_exact_err restates the reference's float/double mix op by op: bit-exact on the host."""
    rng = np.random.default_rng(14)
    r = np.array([0.1, -0.3, 0.2])
    t = np.array([0.2, -0.1, 3.0])
    R = oracle.rodrigues_vec2mat(r)
    pts = rng.uniform(-2, 2, size=(500, 3)).astype(np.float32)
    uv = oracle.project(r, t, FX, FY, CX, CY, pts)
    for i in range(500):
        px, py = np.float32(8 * (i % 80) + 4), np.float32(8 * (i // 80) + 4)
        got = probe.probe_exact_err(_p(np.ascontiguousarray(R)), _p(t), FX, FY, CX, CY, float(pts[i, 0]), float(pts[i, 1]),
                                    float(pts[i, 2]), float(px), float(py))
        dx, dy = np.float32(px - uv[i, 0]), np.float32(py - uv[i, 1])
        want = np.float32(np.sqrt(np.float64(dx) * np.float64(dx) + np.float64(dy) * np.float64(dy)))
        assert np.float32(got) == want


def test_lm_normal_equations_match_numeric_jacobian(oracle, probe):
    """Twist-space sums + chain rule (lm_math.hpp) == J^T J, J^T e of the (rvec,tvec) parametrisation."""
    rng = np.random.default_rng(15)
    for trial in range(5):
        n = 60
        obj = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
        obj[:, 2] += 5
        pose = np.concatenate([rng.normal(size=3) * 0.3, rng.uniform(-0.3, 0.3, 3)])
        img = (oracle.project(pose[:3], pose[3:], FX, FY, CX, CY, obj) + rng.normal(0, 1, (n, 2))).astype(np.float32)
        U, g, e2 = np.zeros(21), np.zeros(6), np.zeros(1)
        probe.probe_lm_normal(_p(obj), _p(img), n, _p(pose), FX, FY, CX, CY, _p(U), _p(g), _p(e2))

        def proj(q):
            R = oracle.rodrigues_vec2mat(q[:3])
            X = (R @ obj.astype(np.float64).T).T + q[3:]
            return np.stack([FX * X[:, 0] / X[:, 2] + CX, FY * X[:, 1] / X[:, 2] + CY], 1).reshape(-1)
        r0 = proj(pose) - img.astype(np.float64).reshape(-1)
        J = np.zeros((2 * n, 6))
        for k in range(6):
            dq = np.zeros(6)
            dq[k] = 1e-6
            J[:, k] = (proj(pose + dq) - proj(pose - dq)) / 2e-6
        Um = np.zeros((6, 6))
        k = 0
        for i in range(6):
            for j in range(i, 6):
                Um[i, j] = Um[j, i] = U[k]
                k += 1
        JtJ = J.T @ J
        assert np.abs(Um - JtJ).max() / np.abs(JtJ).max() < 1e-8
        assert np.abs(g - J.T @ r0).max() / np.abs(J.T @ r0).max() < 1e-8
        assert abs(e2[0] - r0 @ r0) < 1e-9 * (r0 @ r0)
        # the route the kernels take (closed-form chain, monomial moments, software-pipelined pairs with a 0/1 weight
        # on the ragged tail) gives the same normal equations as the entry-by-entry accumulation, for odd n too
        for nn in (n, n - 1, 1):
            U1, g1, e1 = np.zeros(21), np.zeros(6), np.zeros(1)
            U2, g2, e2b = np.zeros(21), np.zeros(6), np.zeros(1)
            probe.probe_lm_normal(_p(obj), _p(img), nn, _p(pose), FX, FY, CX, CY, _p(U1), _p(g1), _p(e1))
            probe.probe_lm_normal_moments(_p(obj), _p(img), nn, _p(pose), FX, CX, CY, _p(U2), _p(g2), _p(e2b))
            np.testing.assert_allclose(U2, U1, rtol=0, atol=1e-11 * np.abs(U1).max())
            np.testing.assert_allclose(g2, g1, rtol=0, atol=1e-11 * np.abs(g1).max())
            assert abs(e2b[0] - e1[0]) <= 1e-12 * e1[0]
        # damped solve == numpy
        for lam in (1e-3, 1.0):
            dx = np.zeros(6)
            probe.probe_lm_solve6(_p(U), _p(g), lam, _p(dx))
            A = Um.copy()
            A[np.diag_indices(6)] *= 1 + lam
            np.testing.assert_allclose(dx, np.linalg.solve(A, g), rtol=1e-8, atol=1e-12)


def test_lm_transform_with_the_translation_folded_in(probe):
    """lm_transform_t (esac_refine_team.hip's route: K = [t]x Mw never formed, B symmetric) == lm_transform on random
    twist-space sums with the structure the kernels produce (entry (3,4) of the normal matrix zero, acc[14] = 0)."""
    rng = np.random.default_rng(21)
    worst = 0.0
    for it in range(200):
        acc = rng.normal(size=27) * 10.0 ** rng.uniform(-2, 4)
        acc[14] = 0.0
        pose = np.concatenate([rng.normal(size=3) * [1e-9, 0.3, 1.5][it % 3], rng.normal(size=3) * [0.1, 3.0][it % 2]])
        worst = max(worst, probe.probe_transform_t_diff(_p(acc), _p(pose)))
    assert worst < 1e-13, worst


def test_lane_dealt_lm_step_equals_the_uniform_route(probe):
    """lm_lanes.hpp (the team kernel's serial section dealt to the lanes of a DPP row: per-lane gather of the totals,
    chain-rule columns, two 3x3-block products on row_newbcast FMAs, Gauss-Jordan over the lanes) on the host's 16-lane
    emulation == lm_moments_to_acc + lm_transform + lm_solve6 on the same totals: the system to rounding, the step to
    the conditioning of the solve, the pivot verdict on a rank-deficient system."""
    rng = np.random.default_rng(33)
    worst_sys = worst_dx = 0.0
    for it in range(300):
        # moments of a real point set, so that the normal matrix is SPD with the structure the kernels produce
        n = [6, 40, 400][it % 3]
        x, y = rng.normal(size=n) * 0.4, rng.normal(size=n) * 0.3
        iz = 1.0 / rng.uniform(1.0, 6.0, size=n)
        ex, ey = rng.normal(size=n) * 3.0, rng.normal(size=n) * 3.0
        xx, yy, xy = x * x, y * y, x * y
        r2, ox, oy = xx + yy, 1 + xx, 1 + yy
        qq, p1, p2, iz2 = 1 + r2, x * iz, y * iz, iz * iz
        mom = [x, y, r2, iz2, iz2 * x, iz2 * y, iz2 * r2, p1, p2, xy * iz, oy * iz, ox * iz, p2 * qq, p1 * qq, xy * (1 + qq),
               xy * xy + oy * oy, xy * xy + ox * ox, xy * ex + oy * ey, ox * ex + xy * ey, x * ey - y * ex, iz * ex, iz * ey,
               p1 * ex + p2 * ey, ex * ex + ey * ey]
        sums = np.array([m.sum() for m in mom] + [1.0, 2.0, float(n)])
        pose = np.concatenate([rng.normal(size=3) * [1e-9, 0.3, 1.5][it % 3], rng.normal(size=3) * [0.1, 3.0][it % 2]])
        lam = 10.0 ** rng.integers(-6, 3)
        U, g, dx = np.zeros(21), np.zeros(6), np.zeros(6)
        ok = probe.probe_lane_step(_p(sums), _p(pose), float(lam), _p(U), _p(g), _p(dx))
        # the uniform route on the same totals
        acc = np.zeros(27)
        f2m = {0: (15, 1), 1: (14, -1), 2: (0, -1), 3: (9, -1), 4: (10, -1), 5: (12, 1), 6: (16, 1), 7: (1, -1), 8: (11, 1), 9: (9, 1),
               10: (13, -1), 11: (2, 1), 12: (8, -1), 13: (7, 1), 15: (3, 1), 16: (4, -1), 17: (3, 1), 18: (5, -1), 19: (6, 1),
               20: (17, -1), 21: (18, 1), 22: (19, 1), 23: (20, 1), 24: (21, 1), 25: (22, -1)}
        for a, (m, sg) in f2m.items():
            acc[a] = sg * sums[m]
        R, Mw, K = np.zeros(9), np.zeros(9), np.zeros(9)
        probe.probe_pose_chain(_p(pose), _p(R), _p(Mw), _p(K))
        Mw, K = Mw.reshape(3, 3), K.reshape(3, 3)
        A6 = np.zeros((6, 6))
        A6[:3, :3] = [[acc[0], acc[1], acc[2]], [acc[1], acc[6], acc[7]], [acc[2], acc[7], acc[11]]]
        A6[:3, 3:] = [[acc[3], acc[4], acc[5]], [acc[8], acc[9], acc[10]], [acc[12], acc[13], acc[14]]]
        A6[3:, :3] = A6[:3, 3:].T
        A6[3:, 3:] = [[acc[15], 0, acc[16]], [0, acc[17], acc[18]], [acc[16], acc[18], acc[19]]]
        M6 = np.block([[Mw, np.zeros((3, 3))], [K, np.eye(3)]])
        Uref = M6.T @ A6 @ M6
        gref = M6.T @ acc[20:26]
        Um = np.zeros((6, 6))
        Um[np.triu_indices(6)] = U
        scale = np.abs(Uref).max()
        worst_sys = max(worst_sys, np.abs(np.triu(Um) - np.triu(Uref)).max() / scale, np.abs(g - gref).max() / np.abs(gref).max())
        Ad = Uref.copy()
        Ad[np.diag_indices(6)] *= 1 + lam
        ref = np.linalg.solve(Ad, gref)
        assert ok == 1
        worst_dx = max(worst_dx, np.abs(dx - ref).max() / np.abs(ref).max() / max(1.0, np.linalg.cond(Ad) * 1e-9))
    assert worst_sys < 1e-13, worst_sys
    assert worst_dx < 1e-8, worst_dx
    # rank-deficient: all points on one viewing ray -> a pivot collapses once lambda is tiny: the verdict must be "not ok"
    sums = np.zeros(27)
    sums[[3, 23]] = [4.0, 1.0]
    U, g, dx = np.zeros(21), np.zeros(6), np.zeros(6)
    assert probe.probe_lane_step(_p(sums), _p(np.array([0.1, 0.2, 0.3, 0.0, 0.0, 1.0])), 1e-16, _p(U), _p(g), _p(dx)) == 0


# ---------------------------------------------------------------- training path (bwd_math.hpp)
def _random_pose(rng, scale=1.0):
    return np.concatenate([rng.normal(size=3) * 0.6 * scale, rng.normal(size=3) * 2.0])


def _gt_from_pose(oracle, pose, rng, noise=0.05):
    """float32 camera transform near pose^-1, as the data loader hands it to esac_backward."""
    p = pose + np.concatenate([rng.normal(size=3) * noise, rng.normal(size=3) * noise])
    return oracle.pose2trans(p).astype(np.float32)


def test_pose_loss_matches_oracle(oracle, probe):
    rng = np.random.default_rng(11)
    worst = 0.0
    for it in range(200):
        pose = _random_pose(rng)
        gt = _gt_from_pose(oracle, pose, rng, noise=[1e-4, 0.05, 1.0][it % 3]).astype(np.float64)
        for cut in (100.0, 0.5):  # with and without the soft clamp
            a = probe.probe_pose_loss(_p(pose), _p(gt), 1.0, 100.0, cut)
            b = oracle.pose_loss(pose, gt, 1.0, 100.0, cut)
            worst = max(worst, abs(a - b) / max(abs(b), 1e-12))
    # rigid inverse (device) vs LU inverse (reference): rounding-level differences, amplified by acos near 0 degrees
    assert worst < 1e-8, worst


def test_pose_dloss_matches_oracle_and_finite_differences(oracle, probe):
    rng = np.random.default_rng(12)
    for it in range(100):
        pose = _random_pose(rng)
        gt_T = _gt_from_pose(oracle, pose, rng, noise=0.05).astype(np.float64)
        gt_pose = oracle.trans2pose(gt_T)
        for cut in (1e9, 0.5):
            a = np.zeros(6)
            probe.probe_pose_dloss(_p(pose), _p(gt_pose), 1.0, 100.0, cut, _p(a))
            b = oracle.pose_dloss(pose, gt_pose, 1.0, 100.0, cut)
            np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-11)
        # without the clamp dLoss is the true gradient of loss() -- central differences agree
        fd = np.zeros(6)
        for k in range(6):
            e = np.zeros(6)
            e[k] = 1e-6
            fd[k] = (oracle.pose_loss(pose + e, gt_T, 1.0, 100.0, 1e9) - oracle.pose_loss(pose - e, gt_T, 1.0, 100.0, 1e9)) / 2e-6
        a = np.zeros(6)
        probe.probe_pose_dloss(_p(pose), _p(gt_pose), 1.0, 100.0, 1e9, _p(a))
        np.testing.assert_allclose(a, fd, rtol=2e-3, atol=2e-3)


def test_trans2pose_orthonormalises_float_pose(oracle):
    rng = np.random.default_rng(13)
    for _ in range(20):
        pose = _random_pose(rng)
        T32 = oracle.pose2trans(pose).astype(np.float32).astype(np.float64)
        back = oracle.trans2pose(T32)
        # float32 rounding of the matrix moves the pose by ~1e-7; the re-orthonormalised result is a proper pose
        np.testing.assert_allclose(back, pose, rtol=0, atol=5e-6)
        R = oracle.rodrigues_vec2mat(back[:3])
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-14)
        # and the SVD variant equals the plain one on an exactly orthonormal input
        Rp = oracle.rodrigues_vec2mat(pose[:3])
        np.testing.assert_allclose(oracle.rodrigues_mat2vec_svd(Rp), oracle.rodrigues_mat2vec(Rp), atol=1e-14)


def test_cell_jacobians_match_oracle(oracle, probe):
    rng = np.random.default_rng(14)
    f32 = np.float32
    n_skip = 0
    for it in range(500):
        pose = _random_pose(rng, 0.3)
        obj = (rng.normal(size=3) * 1.5 + np.array([0, 0, 4.0])).astype(f32)
        # pixel near the projection so that a good share falls inside maxReproj = 100
        uv = oracle.project(pose[:3], pose[3:], FX, FY, CX, CY, obj[None])[0]
        pt = (uv + rng.normal(size=2) * [1.0, 50.0, 200.0][it % 3]).astype(f32)
        ok_a, row_b = oracle.norm_jac_row(pose[:3], pose[3:], 525.0, 320.0, 240.0, obj, pt, 100.0)
        row_a = np.zeros(6)
        ok_p = probe.probe_norm_jac_row(_p(pose[:3].copy()), _p(pose[3:].copy()), 525.0, 320.0, 240.0, obj[0], obj[1], obj[2],
                                        pt[0], pt[1], 100.0, _p(row_a))
        assert bool(ok_p) == bool(ok_a)
        np.testing.assert_array_equal(row_a, row_b)  # same operations in the same order: bit-identical
        n_skip += not ok_a
        d_b = oracle.dproject_dobj(pt, obj, pose[:3], pose[3:], 525.0, 320.0, 240.0, 100.0)
        d_a = np.zeros(3)
        probe.probe_dproject_dobj(pt[0], pt[1], obj[0], obj[1], obj[2], _p(pose[:3].copy()), _p(pose[3:].copy()), 525.0, 320.0,
                                  240.0, 100.0, _p(d_a))
        np.testing.assert_array_equal(d_a, d_b)
    assert 0 < n_skip < 500  # both branches of the maxReproj test were exercised


def test_inv_spd6_equals_reference_pseudo_inverse_on_full_rank(oracle, probe):
    rng = np.random.default_rng(15)
    iu = np.triu_indices(6)
    for it in range(50):
        J = rng.normal(size=(40, 6)) * np.array([100, 100, 100, 30, 30, 10.0])
        A = J.T @ J
        out = np.zeros((6, 6))
        assert probe.probe_inv_spd6(_p(np.ascontiguousarray(A[iu])), _p(out)) == 1
        ref = oracle.pinv_sym6(A)
        np.testing.assert_allclose(out, ref, rtol=1e-8, atol=1e-14)
    # rank deficient / badly conditioned -> reported; the caller then takes the reference's own route, the Jacobi
    # pseudo-inverse, which is the oracle's routine operation for operation
    for trial in range(20):
        J = rng.normal(size=(40, 5)) * 30
        J = np.concatenate([J, J[:, :1] * (1 + (1e-9 if trial % 2 else 0) * rng.normal(size=(40, 1)))], axis=1)
        if trial % 5 == 4:
            J = rng.normal(size=(4, 6)) * 30  # fewer rows than parameters
        A = J.T @ J
        out = np.zeros((6, 6))
        assert probe.probe_inv_spd6(_p(np.ascontiguousarray(A[iu])), _p(out)) == 0
        probe.probe_pinv_sym6(_p(np.ascontiguousarray(A[iu])), _p(out))
        ref = oracle.pinv_sym6(A)
        np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-12 * np.abs(ref).max())
    # and on a well-conditioned matrix it is simply the inverse
    J = rng.normal(size=(40, 6))
    A = J.T @ J
    probe.probe_pinv_sym6(_p(np.ascontiguousarray(A[iu])), _p(out))
    np.testing.assert_allclose(out @ A, np.eye(6), atol=1e-10)


def test_closed_form_lm_chain_equals_the_rodrigues_jacobian_chain(probe):
    """lm_pose_chain (left Jacobian of SO(3) in closed form) == lm_chain(dR/drvec): same R, Mw, K -- to rounding for
    ordinary rotations (up to beyond pi); for tiny rotations, where 1 - cos(theta) cancels in the Rodrigues-Jacobian
    route, the closed form is checked against the power series of the left Jacobian instead."""
    rng = np.random.default_rng(21)
    for scale, bar in ((2e-2, 1e-11), (0.5, 1e-13), (3.0, 1e-13), (3.2, 1e-13)):
        worst = 0.0
        for it in range(400):
            pose = np.concatenate([rng.normal(size=3) * scale, rng.normal(size=3) * 3.0])
            worst = max(worst, probe.probe_chain_diff(_p(pose)))
        assert worst < bar, (scale, worst)
    assert probe.probe_chain_diff(_p(np.array([0, 0, 0, 1.0, -2.0, 0.5]))) == 0.0

    def skew(v):
        return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])
    for scale in (1e-12, 1e-9, 1e-6, 1e-4, 3e-3, 6e-3, 1e-2):
        for it in range(50):
            pose = np.concatenate([rng.normal(size=3) * scale, rng.normal(size=3) * 3.0])
            R, Mw, K = np.zeros(9), np.zeros(9), np.zeros(9)
            probe.probe_pose_chain(_p(pose), _p(R), _p(Mw), _p(K))
            S1 = skew(pose[:3])
            S2 = S1 @ S1
            th2 = float(pose[:3] @ pose[:3])
            # series of the left Jacobian: I + (1-cos)/th^2 [r]x + (th-sin)/th^3 [r]x^2
            Jl = (np.eye(3) + (0.5 - th2 / 24 + th2 ** 2 / 720 - th2 ** 3 / 40320) * S1
                  + (1 / 6 - th2 / 120 + th2 ** 2 / 5040 - th2 ** 3 / 362880) * S2)
            np.testing.assert_allclose(Mw.reshape(3, 3), Jl, rtol=0, atol=2e-14)
            np.testing.assert_allclose(K.reshape(3, 3), skew(pose[3:]) @ Jl, rtol=0, atol=5e-13)


def test_pose_rotation_series_against_high_precision(probe):
    """lm_pose_rotation / lm_pose_left_jacobian (lm_math.hpp: Taylor series at a quarter of the angle, two doublings,
    C = (1 - A) / x) against 50-digit arithmetic over the whole series range |r|^2 <= 10 -- up to angle pi and beyond --
    and on the trigonometric route past it: R and the left Jacobian of SO(3) to a few ulp of their largest entries."""
    mpmath = pytest.importorskip("mpmath")
    mpmath.mp.dps = 50
    rng = np.random.default_rng(5)
    worst_R = worst_J = 0.0
    angles = np.concatenate([rng.uniform(0.0, np.sqrt(10.0), 300), [1e-7, 1e-3, np.pi, 3.1622, 3.1624, 3.5, 6.0]])
    for th in angles:
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        pose = np.concatenate([axis * th, rng.normal(size=3)])
        R, Mw, K = np.zeros(9), np.zeros(9), np.zeros(9)
        probe.probe_pose_chain(_p(pose), _p(R), _p(Mw), _p(K))
        r = [mpmath.mpf(float(v)) for v in pose[:3]]
        x = r[0] * r[0] + r[1] * r[1] + r[2] * r[2]
        t = mpmath.sqrt(x)
        A, B, C = mpmath.sin(t) / t, (1 - mpmath.cos(t)) / x, (t - mpmath.sin(t)) / (t * x)
        S = mpmath.matrix([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
        Rr = mpmath.eye(3) + A * S + B * (S * S)
        Jl = mpmath.eye(3) + B * S + C * (S * S)
        for i in range(3):
            for j in range(3):
                worst_R = max(worst_R, abs(float(mpmath.mpf(float(R[3 * i + j])) - Rr[i, j])))
                worst_J = max(worst_J, abs(float(mpmath.mpf(float(Mw[3 * i + j])) - Jl[i, j])))
    assert worst_R < 3e-15, worst_R
    assert worst_J < 3e-15, worst_J


def test_fp32_screen_never_rejects_an_accepted_try():
    """The one-sided contract of the sampling screen (esac_amd/csrc/p3p_screen.hpp: screen_setup + p3p_screen_roots, as
    k_sample_prescreen / k_sample_screened run it): on the kernels' own source compiled for the host, over 1.5e6 random tries on true-expert, garbage-expert, noise-free,
    coarse-grid and far-from-origin maps, no try the fp64 route ACCEPTS is ruled out by the screen at HALF the margin the
    kernel uses, the screen's error of every accepted try stays within 0.1 px of tau, and on a garbage map it clears
    more than 99 % of the tries (which is the point of it)."""
    import ctypes as C
    from esac_amd import synthetic as S
    from tests.native import build as nb
    lib = C.CDLL(nb.build_screen_probe())
    margins = np.array([1.5, 3.0], np.float32)  # the kernel: SCREEN_MARGIN = 3 px

    def run(coords, f, n, seed, mode=3):  # mode 3: the fp64-root screen as the kernels run it
        out = np.zeros(40)
        c = np.ascontiguousarray(coords, np.float32)
        _, H, W = c.shape
        lib.probe_screen(c.ctypes.data_as(C.c_void_p), H, W, f["sub"], f["shift"][0], f["shift"][1], C.c_float(f["focal"]),
                         C.c_float(f["ppx"]), C.c_float(f["ppy"]), C.c_float(10.0), C.c_uint64(seed), C.c_longlong(n),
                         margins.ctypes.data_as(C.c_void_p), 2, mode, out.ctypes.data_as(C.c_void_p))
        return out

    f = S.make_frame(3, E=2, true_expert=0)
    off = np.array([1200.0, -800.0, 950.0], np.float32)[:, None, None]
    cases = [("true expert", f["coords"][0], f), ("garbage expert", f["coords"][1], f),
             ("noise-free", S.make_frame(7, noise=0.0, outlier_frac=0.0)["coords"][0], f),
             ("coarse grid", None, S.make_frame(8, H=24, W=32, sub=20)), ("far from the origin", f["coords"][0] + off, f)]
    for k, (name, coords, fr) in enumerate(cases):
        out = run(fr["coords"][0] if coords is None else coords, fr, 300000, 40 + k)
        tries, accepted = out[0], out[1]
        assert out[12] == 0 and out[13] == 0, (name, out[12:14])      # fp64-accepted tries the screen would have rejected
        assert accepted > 0 and out[3] <= 10.0 + 0.1, (name, out[3])  # largest screen error among accepted tries
        if name == "garbage expert":
            assert out[5] / tries < 0.01, out[5] / tries              # "maybe" fraction at the kernel's margin


def test_screen_defers_when_the_fourth_point_sits_at_the_camera_centre():
    """The one false reject a 3.6e10-try host campaign found (scripts/dev/screen_adversarial.py, map "plane
    warped, x and y quantised"): the 4th cell repeats a base point's scene coordinates 8 px away in the image, the solution
    puts the camera centre 3.5 mm from it, and the alignment-free evaluation of the screen (14.0 px) and the fp64 route's
    least-squares alignment (9.98 px: accepted) settle the triangle's residual mismatch differently.  The screen must say
    "maybe" there -- on its private copy of the roots and on the exact route's own."""
    import ctypes as C
    from tests.native import build as nb
    lib = C.CDLL(nb.build_screen_probe())
    pts = np.array([[1.0, -0.0, 2.0], [0.0, 0.25, 2.0], [1.75, 0.5, 2.0], [1.75, 0.5, 2.0]], np.float32)
    px = np.array([[476, 188], [340, 388], [612, 412], [612, 420]], np.float32)
    out = np.zeros(4)
    lib.probe_one_values(pts.ctypes.data_as(C.c_void_p), px.ctypes.data_as(C.c_void_p), C.c_float(525.0), C.c_float(320.0),
                         C.c_float(240.0), C.c_float(10.0), out.ctypes.data_as(C.c_void_p))
    assert out[3] == 1.0 and 9.9 < out[2] < 10.0, out   # the fp64 route accepts, barely
    assert out[0] == -1.0 and out[1] == -1.0, out       # ESAC_SCREEN_MAYBE on both sets of roots
    # the same triangle seen from a metre away is none of the guard's business
    px2 = np.array([[476, 188], [340, 388], [612, 412], [100, 80]], np.float32)
    pts2 = pts.copy()
    pts2[3] = [-1.5, -1.0, 2.5]
    lib.probe_one_values(pts2.ctypes.data_as(C.c_void_p), px2.ctypes.data_as(C.c_void_p), C.c_float(525.0), C.c_float(320.0),
                         C.c_float(240.0), C.c_float(10.0), out.ctypes.data_as(C.c_void_p))
    assert out[0] > 13.0 and out[3] == 0.0, out         # a plain rejection, decided by the screen itself


def test_fast_quartic_agrees_with_the_exact_route_or_says_maybe():
    """quartic_roots_fast (the sampling screen's copy of the Ferrari solve: contracted arithmetic, fp32-seeded Newton cubic
    root) next to quartic_real_roots (the route the decision uses) on quartics BUILT to be hard: a root pair closing from 1e-1
    to 1e-13, a complex pair whose imaginary part shrinks to nothing (real roots about to appear), triple clusters, and both
    at once -- the cases in which Ferrari's root count is rounding.  Contract: the fast copy either reports -1 ("not
    reproducible": the screen then lets the try through) or returns the exact route's number of roots with the same values.
    A silent disagreement is a dropped hypothesis on the GPU (found by calibration in round 2: ~6 per million accepted tries)."""
    import ctypes as C
    from tests.native import build as nb
    lib = C.CDLL(nb.build_screen_probe())
    lib.probe_quartic.argtypes = [C.c_double] * 5 + [C.c_void_p]
    rng = np.random.default_rng(7)

    def coeffs(roots_real, pairs):  # monic quartic from real roots and complex pairs (re, im)
        p = np.poly1d([1.0])
        for r in roots_real:
            p *= np.poly1d([1.0, -r])
        for re, im in pairs:
            p *= np.poly1d([1.0, -2 * re, re * re + im * im])
        return p.coeffs

    cases = []
    for _ in range(4000):
        base = rng.uniform(0.2, 3.0, 4) * rng.choice([-1, 1], 4)
        gap = 10.0 ** rng.uniform(-13, -1)
        kind = rng.integers(0, 5)
        if kind == 0:    # a closing real pair + two ordinary real roots
            cases.append(coeffs([base[0], base[0] + gap, base[1], base[2]], []))
        elif kind == 1:  # a complex pair about to become real + two real roots
            cases.append(coeffs([base[1], base[2]], [(base[0], gap)]))
        elif kind == 2:  # a closing pair next to a complex pair
            cases.append(coeffs([base[0], base[0] + gap], [(base[1], abs(base[2]))]))
        elif kind == 3:  # a cluster of three
            cases.append(coeffs([base[0], base[0] + gap, base[0] - 0.7 * gap, base[1]], []))
        else:            # two closing pairs
            cases.append(coeffs([base[0], base[0] + gap, base[1], base[1] + 0.5 * gap], []))
    said_maybe = agreed = 0
    out = np.zeros(10)
    for c in cases:
        scale = 10.0 ** rng.uniform(-3, 3)  # the P3P quartic is not monic
        lib.probe_quartic(*(float(v * scale) for v in c), out.ctypes.data_as(C.c_void_p))
        n_exact, n_fast = int(out[0]), int(out[5])
        if n_fast < 0:
            said_maybe += 1
            continue
        assert n_fast == n_exact, (c, n_exact, n_fast, out)
        xe, xf = np.sort(out[1:1 + n_exact]), np.sort(out[6:6 + n_fast])
        assert np.allclose(xe, xf, rtol=1e-6, atol=1e-9), (c, xe, xf)
        agreed += 1
    assert agreed > len(cases) // 4 and said_maybe > 0, (agreed, said_maybe)  # it is a guard, not a blanket refusal



def test_ill_conditioned_minimal_set_is_a_known_divergence(oracle, probe):
    """Round 6's 1000-frame sweep with several experts (profiles/r06_sweep_1000_several_experts.txt) found ONE hypothesis whose
    accepted try differs between the oracle and the HIP path -- on every route, the guaranteed fp64 one included: frame 5874 of
    the synthetic generator, hypothesis 1746 on the garbage map of a wrong expert, try 2198: four cells of a 3 x 2 pixel-cell
    neighbourhood (an image triangle 16 pixels wide) whose scene points lie ~2 m apart, i.e. a camera ~70 m away.  Three-point
    pose from such a sliver is ill-conditioned beyond what either formulation resolves: NEITHER solver's pose reprojects the
    three base points exactly (5-10 px off, where a well-posed sample gives < 1e-6), the two poses differ by metres, and whether
    the four errors stay below tau = 10 px is decided by rounding -- the oracle (Gao + Ferrari + Horn) says yes at 6.5 px, the
    device formulation (Gao + Ferrari + triad / Newton) no at 10.55 px.  What OpenCV's own P3P would say cannot be known here
    (OpenCV internals unpinned).  The hypothesis is a wrong-expert straggler that does not win; winner, pose and
    every other hypothesis of that frame agree.  This test pins the sample so that a change of either solver that moves it
    is noticed."""
    f = S.make_frame(5874, E=12, true_expert=874 % 12)
    cells = [(57, 47), (58, 48), (56, 48), (56, 47)]
    obj = np.array([[f["coords"][7, c, y, x] for c in range(3)] for x, y in cells], np.float64)
    img = np.array([[x * 8 + 4, y * 8 + 4] for x, y in cells], np.float64)

    def reproj(r, t):
        th = np.linalg.norm(r)
        K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]]) / th
        R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
        P = obj @ R.T + t
        return np.hypot(FX * P[:, 0] / P[:, 2] + CX - img[:, 0], FY * P[:, 1] / P[:, 2] + CY - img[:, 1])

    ok_o, r_o, t_o = oracle.p3p(obj, img, FX, FY, CX, CY)
    ok_p, r_p, t_p = _probe_p3p(probe, obj, img)
    assert ok_o and ok_p
    e_o, e_p = reproj(r_o, t_o), reproj(r_p, t_p)
    assert np.abs(t_o - t_p).max() > 1.0                  # metres apart
    assert e_o[:3].min() > 1.0 and e_p[:3].min() > 1.0    # neither reprojects its own three base points: not a resolved P3P solution
    assert e_o.max() < 10.0 < e_p.max()                   # ... and tau = 10 px falls between the two
    assert t_o[2] > 50 and t_p[2] > 50                    # the sliver's camera: tens of metres out
