"""tests/golden/ref_*.npz were produced by the REFERENCE's own esac_forward / esac_backward (compiled from
/root/reference by oracle/_ref, see tests/golden/make_ref_golden.py) together with the mt19937 draws it consumed.
The oracle, fed the recorded draws through its callback RNG, must reproduce the reference's outputs BIT FOR BIT --
a pin of the oracle that does not need /root/reference at test time (GPU box, CI)."""
import glob
import os

import numpy as np
import pytest

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_*.npz")))


def _replayer(draws):
    it = iter(draws)

    def irand(lo, hi):
        rlo, rhi, v = next(it)
        assert (lo, hi) == (int(rlo), int(rhi)), "the oracle asked for a different draw than the reference made"
        return int(v)
    return irand, it


def _kw(g):
    kw = dict(shift_x=int(g["shift"][0]), shift_y=int(g["shift"][1]), focal=float(g["focal"]), ppx=float(g["ppx"]),
              ppy=float(g["ppy"]), sub_sampling=int(g["sub"]))
    for k in ("inlier_thresh", "inlier_alpha", "inlier_beta", "max_reproj"):
        if k in g.files:
            kw[k] = float(g[k])
    return kw


def test_reference_fixtures_exist():
    kinds = sorted(str(np.load(p)["kind"]) for p in FIXTURES)
    assert kinds.count("forward") >= 3 and kinds.count("backward") >= 3


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_reproduces_reference_outputs(oracle, path):
    g = np.load(path)
    irand, it = _replayer(g["draws"])
    if str(g["kind"]) == "forward":
        o = oracle.forward(g["coords"], g["assign"], irand=irand, **_kw(g))
        np.testing.assert_array_equal(o["sample_xy"], g["ref_sample_xy"])
        np.testing.assert_array_equal(o["hyps"], g["ref_hyps"])
        np.testing.assert_array_equal(o["scores"], g["ref_scores"])
        assert o["winner"] == int(g["ref_winner"]) and o["expert"] == int(g["ref_expert"])
        np.testing.assert_array_equal(o["refined"], g["ref_refined"])
        np.testing.assert_array_equal(o["inlier_map"], g["ref_inlier_map"])
        np.testing.assert_array_equal(o["pose"], g["ref_pose"])  # what esac_forward wrote into outPose
    else:
        grads = np.zeros_like(g["coords"])
        o = oracle.backward(g["coords"], grads, g["assign"], g["gt_pose"], w_rot=float(g["w_rot"]), w_trans=float(g["w_trans"]),
                            loss_cut=float(g["loss_cut"]), irand=irand, **_kw(g))
        assert o["loss"] == float(g["ref_loss"])                  # the value esac_backward returned
        np.testing.assert_array_equal(grads, g["ref_gradients"])  # the tensor it accumulated into
        assert np.abs(grads).max() > 0
    assert next(it, None) is None, "the oracle consumed fewer draws than the reference"
