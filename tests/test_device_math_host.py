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
        # damped solve == numpy
        for lam in (1e-3, 1.0):
            dx = np.zeros(6)
            probe.probe_lm_solve6(_p(U), _p(g), lam, _p(dx))
            A = Um.copy()
            A[np.diag_indices(6)] *= 1 + lam
            np.testing.assert_allclose(dx, np.linalg.solve(A, g), rtol=1e-8, atol=1e-12)
