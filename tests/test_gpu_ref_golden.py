"""HIP path vs outputs of the REFERENCE ITSELF: tests/golden/ref_fwd_*.npz hold what the reference's own esac_forward
(compiled from /root/reference by oracle/_ref, see tests/golden/make_ref_golden.py) produced -- its hypotheses, scores,
winner, inlier map, refined pose, returned pose.  The reference's hypotheses are handed to the device
(esac_hip_write_hyps); scoring, selection and refinement then run on the GPU and must reproduce the reference's
numbers directly, without the oracle in between (the sampling stage cannot be replayed: mt19937 vs Philox)."""
import glob
import os

import numpy as np
import pytest
import torch

from esac_amd import api
from esac_amd import synthetic as S

pytestmark = pytest.mark.gpu

FIXTURES = sorted(p for p in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_fwd_*.npz")))


def _setup(engine, g):
    sc = torch.from_numpy(g["coords"]).cuda()
    ha = torch.from_numpy(g["assign"]).cuda()
    E, _, H, W = g["coords"].shape
    p = engine.make_params(E, H, W, len(g["assign"]), shift_x=int(g["shift"][0]), shift_y=int(g["shift"][1]),
                           focal=float(g["focal"]), ppx=float(g["ppx"]), ppy=float(g["ppy"]), sub_sampling=int(g["sub"]),
                           inlier_thresh=float(g["inlier_thresh"]), inlier_alpha=float(g["inlier_alpha"]),
                           inlier_beta=float(g["inlier_beta"]), max_reproj=float(g["max_reproj"]))
    return sc, ha, p


def _check_tail(engine, g):
    res = engine.read(api.BUF_RESULT)
    assert int(res[api.RES_HYP]) == int(g["ref_winner"]) and int(res[api.RES_EXPERT]) == int(g["ref_expert"])
    np.testing.assert_array_equal(engine.read(api.BUF_INLIER_MAP), g["ref_inlier_map"])
    np.testing.assert_allclose(res[api.RES_RVEC:api.RES_RVEC + 6], g["ref_refined"], rtol=0, atol=1e-6)
    pose = res[api.RES_POSE:api.RES_POSE + 16].reshape(4, 4)
    np.testing.assert_allclose(pose, g["ref_pose"], rtol=0, atol=2e-6)  # float32 4x4 the reference wrote into outPose
    r, t = S.pose_errors(pose, g["ref_pose"])
    assert r <= 1e-4 and t <= 1e-3
    return res


def test_fixtures_present():
    assert len(FIXTURES) >= 3


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_exact_scores_selection_refinement_match_the_reference(engine, path):
    g = np.load(path)
    sc, ha, p = _setup(engine, g)
    engine.write_hyps(g["ref_hyps"])
    engine.score_exact(sc, ha, p)  # reference arithmetic for every hypothesis (esac_util.h:235-260)
    np.testing.assert_allclose(engine.read(api.BUF_SCORES), g["ref_scores"], rtol=1e-12, atol=1e-11)
    engine.refine(sc, ha, p)       # draw(argmax) + refineHyp + pose2trans
    res = _check_tail(engine, g)
    assert abs(res[api.RES_SCORE] - g["ref_scores"][int(g["ref_winner"])]) <= 1e-10


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_default_path_on_the_reference_hypotheses(engine, path):
    """The shipped route (fp32 streaming score, exact re-score of the contenders) from the reference's hypotheses."""
    g = np.load(path)
    sc, ha, p = _setup(engine, g)
    engine.write_hyps(g["ref_hyps"])
    engine.score(sc, ha, p)
    engine.select(sc, ha, p)
    scores = engine.read(api.BUF_SCORES)
    flags = engine.read(api.BUF_EXACT_FLAGS).astype(bool)
    alpha = float(g["inlier_alpha"])
    assert flags[int(g["ref_winner"])]
    np.testing.assert_allclose(scores[flags], g["ref_scores"][flags], rtol=1e-12, atol=1e-11)
    assert np.abs(scores[~flags] - g["ref_scores"][~flags]).max(initial=0) <= 2e-5 * alpha
    engine.refine(sc, ha, p)
    _check_tail(engine, g)
