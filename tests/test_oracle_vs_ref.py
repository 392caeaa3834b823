"""oracle/ vs oracle/_ref: the reference's OWN sources (esac_util.h, esac_types.h, thread_rand.cpp, compiled from
/root/reference against the OpenCV/ATen stand-in shim) run next to the oracle on the same inputs and the same
mt19937 stream (single thread).  Every stage output must agree BIT FOR BIT: this pins the oracle's restatement of
the reference's control flow and float/double mixes.  (OpenCV's internals stay restated from memory: the shim
forwards them to the oracle's routines.)  Skipped where oracle/_ref was never built."""
import os

import numpy as np
import pytest

from esac_amd import synthetic as S

ref_binding = pytest.importorskip("oracle.ref_binding")
if ref_binding.build() is None or not os.path.exists(ref_binding.LIB_PATH):
    pytest.skip("oracle/_ref not built (reference sources not mounted)", allow_module_level=True)

CASES = [
    dict(k=0, N=64),
    dict(k=1, N=256),
    dict(k=2, N=128, E=3, true_expert=1, mode="gating"),
    dict(k=3, N=48, H=24, W=32, sub=20, shift=(3, -2)),
    dict(k=4, N=64, params=dict(inlier_thresh=6.0, inlier_alpha=50.0, inlier_beta=0.8, max_reproj=60.0)),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "k%d_N%d" % (c["k"], c["N"]))
def test_oracle_matches_reference_sources_bit_for_bit(oracle, case):
    fkw = {k: v for k, v in case.items() if k in ("k", "E", "true_expert", "H", "W", "sub", "shift")}
    f = S.make_frame(**fkw)
    ha = S.gating_assignment(f, case["N"], mode=case.get("mode", "single"))
    kw = dict(shift_x=f["shift"][0], shift_y=f["shift"][1], focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"],
              sub_sampling=f["sub"], **case.get("params", {}))
    ref = ref_binding.forward(f["coords"], ha, seed=1305, **kw)
    ora = oracle.forward(f["coords"], ha, irand=ref_binding.replay_irand(1305), **kw)
    np.testing.assert_array_equal(ora["sample_xy"], ref["sample_xy"])  # sampleHypotheses: same cells, same accepted try
    np.testing.assert_array_equal(ora["hyps"], ref["hyps"])            # safeSolvePnP results as stored by the reference
    np.testing.assert_array_equal(ora["scores"], ref["scores"])        # getReproErrs + getHypScores, float/double mix
    assert ora["winner"] == ref["winner"] and ora["expert"] == ref["expert"]  # softMax + draw(false)
    assert ora["entropy"] == ref["entropy"]
    np.testing.assert_array_equal(ora["refined"], ref["refined"])      # refineHyp: inlier sets, stopping rule, re-fits
    np.testing.assert_array_equal(ora["inlier_map"], ref["inlier_map"])
    np.testing.assert_array_equal(ora["pose"], ref["pose"])            # pose2trans + float cast


def test_budget_exhaustion_matches_reference(oracle):
    """Tries run out: the reference keeps the state of the last try (esac_util.h:154-223)."""
    f = S.make_frame(5)
    ha = S.gating_assignment(f, 32)
    ref = ref_binding.forward(f["coords"], ha, seed=77, max_tries=2)
    ora = oracle.forward(f["coords"], ha, irand=ref_binding.replay_irand(77), max_tries=2)
    assert (ora["tries"] == -1).any()
    np.testing.assert_array_equal(ora["sample_xy"], ref["sample_xy"])
    np.testing.assert_array_equal(ora["hyps"], ref["hyps"])
    np.testing.assert_array_equal(ora["scores"], ref["scores"])
    np.testing.assert_array_equal(ora["pose"], ref["pose"])


def test_reference_esac_forward_itself(oracle):
    """esac_forward compiled from the reference's esac.cpp (torch/OpenCV stand-ins) == oracle, same mt19937 stream."""
    f = S.make_frame(6, E=2, true_expert=1)
    ha = S.gating_assignment(f, 96, mode="gating")
    e, pose = ref_binding.esac_forward(f["coords"], ha, seed=1305)
    ora = oracle.forward(f["coords"], ha, irand=ref_binding.replay_irand(1305))
    assert e == ora["expert"]
    np.testing.assert_array_equal(pose, ora["pose"])


BWD_CASES = [
    dict(k=0, N=64),
    dict(k=1, N=96, E=3, true_expert=2, mode="gating"),
    dict(k=2, N=48, H=24, W=32, sub=20, shift=(3, -2)),
    dict(k=3, N=64, loss=dict(w_rot=2.0, w_trans=50.0, loss_cut=0.5)),   # soft-clamped loss branch (esac_loss.h:79-80,133-137)
    dict(k=4, N=64, params=dict(inlier_thresh=6.0, inlier_alpha=50.0, inlier_beta=0.8, max_reproj=60.0)),
]


@pytest.mark.parametrize("case", BWD_CASES, ids=lambda c: "k%d_N%d" % (c["k"], c["N"]))
def test_oracle_backward_matches_reference_esac_backward(oracle, case):
    """esac_backward compiled from the reference's esac.cpp + esac_loss.h + esac_derivative.h vs the oracle's
    restatement: same expected loss, same gradient tensor (accumulated onto a non-zero start, as `+=` implies)."""
    fkw = {k: v for k, v in case.items() if k in ("k", "E", "true_expert", "H", "W", "sub", "shift")}
    f = S.make_frame(**fkw)
    ha = S.gating_assignment(f, case["N"], mode=case.get("mode", "single"))
    gt = f["gt_pose"].astype(np.float32)
    gt[:3, 3] += np.float32(0.03)  # a ground truth that is not the optimum: non-trivial loss gradient
    kw = dict(shift_x=f["shift"][0], shift_y=f["shift"][1], focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"],
              sub_sampling=f["sub"], **case.get("params", {}), **case.get("loss", {}))
    rng = np.random.default_rng(5)
    g_ref = rng.normal(0, 1e-3, f["coords"].shape).astype(np.float32)
    g_ora = g_ref.copy()
    loss_ref = ref_binding.esac_backward(f["coords"], g_ref, ha, gt, seed=1305, **kw)
    out = oracle.backward(f["coords"], g_ora, ha, gt, irand=ref_binding.replay_irand(1305), **kw)
    assert (out["probs"] >= 1e-3).sum() >= 1
    assert out["loss"] == pytest.approx(loss_ref, rel=1e-12, abs=1e-12)
    assert np.abs(g_ref).max() > 1e-2  # a real gradient came out
    np.testing.assert_allclose(g_ora, g_ref, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf, 1e30])
def test_non_finite_scene_coordinate_matches_reference(oracle, bad):
    """What the REFERENCE does with a scene coordinate that is not finite: `std::min((float)cv::norm(curPt), maxReproj)`
    (esac_util.h:358) returns its FIRST argument when that is NaN, so the cell's error stays NaN, every hypothesis' score is NaN
    (esac_util.h:248-250), softMax yields NaN throughout and draw() keeps index 0 (esac_util.h:519-523: `maxProb < 0` is true
    once, nothing compares greater than NaN afterwards): the reference refines hypothesis 0, whatever it is.  A huge but finite
    coordinate (1e30) clamps to maxReproj and changes nothing.  The oracle restates exactly that (the HIP path deliberately does
    not: DESIGN.md, deviation table; tests/test_gpu_edge.py::test_non_finite_scene_coordinates)."""
    f = S.make_frame(7)
    ha = S.gating_assignment(f, 48)
    coords = f["coords"].copy()
    coords[0, 1, 17, 23] = bad
    kw = dict(focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"])
    ref = ref_binding.forward(coords, ha, seed=1305, **kw)
    ora = oracle.forward(coords, ha, irand=ref_binding.replay_irand(1305), **kw)
    np.testing.assert_array_equal(ora["sample_xy"], ref["sample_xy"])
    np.testing.assert_array_equal(ora["hyps"], ref["hyps"])
    np.testing.assert_array_equal(ora["scores"], ref["scores"])  # (assert_array_equal treats NaN == NaN)
    assert ora["winner"] == ref["winner"]
    np.testing.assert_array_equal(ora["refined"], ref["refined"])
    np.testing.assert_array_equal(ora["inlier_map"], ref["inlier_map"])
    np.testing.assert_array_equal(ora["pose"], ref["pose"])
    if np.isfinite(bad):
        assert np.isfinite(ref["scores"]).all()
    else:
        assert np.isnan(ref["scores"]).all() and ref["winner"] == 0
        assert ref["inlier_map"][17, 23] == 0 and ref["inlier_map"].sum() > 1000  # hypothesis 0 is refined as usual, without that cell
