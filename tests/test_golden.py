"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the CPU oracle).

CPU: the oracle reproduces them (regression pin; discrete outputs bit-exact).
GPU: the HIP path reproduces them through the C ABI (pose within 1e-4 rad / 1e-3 m, index work bit-exact)."""
import glob
import os

import numpy as np
import pytest

from esac_amd import synthetic as S

FIXTURES = sorted(p for p in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz"))
                  if not os.path.basename(p).startswith("ref_"))  # ref_*.npz: tests/test_ref_golden.py


def test_fixtures_exist():
    assert len(FIXTURES) >= 4


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path)
    o = oracle.forward(g["coords"], g["assign"], shift_x=int(g["shift"][0]), shift_y=int(g["shift"][1]),
                       focal=float(g["focal"]), ppx=float(g["ppx"]), ppy=float(g["ppy"]), sub_sampling=int(g["sub"]),
                       seed=int(g["seed"]), call=int(g["call"]))
    np.testing.assert_array_equal(o["sample_xy"], g["sample_xy"])
    np.testing.assert_array_equal(o["tries"], g["tries"])
    assert o["winner"] == int(g["winner"]) and o["expert"] == int(g["expert"])
    assert o["ref_steps"] == int(g["ref_steps"])
    np.testing.assert_array_equal(o["inlier_counts"], g["inlier_counts"])
    np.testing.assert_array_equal(o["inlier_map"], g["inlier_map"])
    np.testing.assert_allclose(o["hyps"], g["hyps"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(o["scores"], g["scores"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(o["refined"], g["refined"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(o["pose"], g["pose"], rtol=0, atol=1e-6)
    r_err, t_err = S.pose_errors(o["pose"], g["gt_pose"])
    assert r_err < np.radians(1.0) and t_err < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_hip_path_reproduces_golden(engine, path):
    import torch
    from esac_amd import api
    g = np.load(path)
    E, _, H, W = g["coords"].shape
    N = len(g["assign"])
    p = engine.make_params(E, H, W, N, shift_x=int(g["shift"][0]), shift_y=int(g["shift"][1]), focal=float(g["focal"]),
                           ppx=float(g["ppx"]), ppy=float(g["ppy"]), sub_sampling=int(g["sub"]), seed=int(g["seed"]),
                           call=int(g["call"]))
    res = engine.forward_device(torch.from_numpy(g["coords"]).cuda(), torch.from_numpy(g["assign"]).cuda(), p)
    np.testing.assert_array_equal(engine.read(api.BUF_SAMPLE_XY), g["sample_xy"])
    np.testing.assert_array_equal(engine.read(api.BUF_TRIES), g["tries"])
    np.testing.assert_allclose(engine.read(api.BUF_HYPS), g["hyps"], rtol=0, atol=1e-6)
    assert int(res[api.RES_HYP]) == int(g["winner"]) and int(res[api.RES_EXPERT]) == int(g["expert"])
    assert int(res[api.RES_REF_STEPS]) == int(g["ref_steps"])
    np.testing.assert_array_equal(engine.read(api.BUF_INLIER_COUNTS), g["inlier_counts"])
    np.testing.assert_array_equal(engine.read(api.BUF_INLIER_MAP), g["inlier_map"])
    r_err, t_err = S.pose_errors(res[api.RES_POSE:api.RES_POSE + 16].reshape(4, 4), g["pose"])
    assert r_err <= 1e-4 and t_err <= 1e-3, (r_err, t_err)
    np.testing.assert_allclose(res[api.RES_SCORE], g["scores"][int(g["winner"])], rtol=0, atol=1e-7)
