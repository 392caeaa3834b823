"""Caller-side glue (esac_amd/harness.py = test_esac.py:135-289 with device-resident tensors)."""
import io
import math

import numpy as np
import pytest
import torch

from esac_amd import harness
from esac_amd import synthetic as S


def test_rodrigues_vector_matches_oracle(oracle):
    rng = np.random.default_rng(0)
    for _ in range(50):
        r = rng.normal(size=3) * rng.uniform(0.01, 3.0)
        R = oracle.rodrigues_vec2mat(r)
        got = harness.rodrigues_vector(R)
        if np.linalg.norm(r) < math.pi:
            np.testing.assert_allclose(got, r, atol=1e-9)
    np.testing.assert_array_equal(harness.rodrigues_vector(np.eye(3)), np.zeros(3))
    Rpi = np.diag([1.0, -1.0, -1.0])  # rotation by pi about x
    np.testing.assert_allclose(np.abs(harness.rodrigues_vector(Rpi)), [math.pi, 0, 0], atol=1e-9)


def test_pose_metrics_and_pose_file_line():
    f = S.make_frame(0)
    gt = f["gt_pose"]
    assert harness.pose_errors_deg_cm(gt, gt) == (0.0, 0.0)
    other = gt.copy()
    other[:3, 3] += [0.03, 0.0, 0.04]
    r, t = harness.pose_errors_deg_cm(other, gt)
    assert r < 1e-9 and abs(t - 5.0) < 1e-9  # cm
    line = harness.pose_file_line("frame-000001", gt)
    parts = line.split()
    assert parts[0] == "frame-000001" and len(parts) == 8
    q = np.array([float(v) for v in parts[1:5]])
    assert abs(np.linalg.norm(q) - 1.0) < 1e-5
    inv = np.linalg.inv(gt)
    np.testing.assert_allclose([float(v) for v in parts[5:8]], inv[:3, 3], atol=1e-5)


class _SyntheticExpert(torch.nn.Module):
    """Stands in for the Expert FCN (code/expert.py): returns the ray-cast map of the current frame."""
    def __init__(self, store, e):
        super().__init__()
        self.store, self.e = store, e

    def forward(self, image):
        return self.store["coords"][self.e:self.e + 1]


class _SyntheticGating(torch.nn.Module):
    def __init__(self, store):
        super().__init__()
        self.store = store

    def forward(self, image):
        return self.store["log_gating"]


@pytest.mark.gpu
def test_evaluation_loop_on_device():
    """Whole loop on synthetic experts: device-resident maps, on-device multinomial / histc, only active experts run."""
    E = 4
    store = {}
    gating = _SyntheticGating(store)
    experts = [_SyntheticExpert(store, e) for e in range(E)]
    gen = torch.Generator(device="cuda").manual_seed(5)

    def samples():
        for k in range(8):
            true_e = k % E
            f = S.make_frame(200 + k, E=E, true_expert=true_e)
            store["coords"] = torch.from_numpy(f["coords"]).cuda()
            logits = torch.full((1, E), -4.0, device="cuda")
            logits[0, true_e] = 4.0
            store["log_gating"] = torch.log_softmax(logits, dim=1)
            yield "img%03d" % k, torch.zeros(1, 3, 480, 640, device="cuda"), f["focal"], f["gt_pose"], true_e

    log = io.StringIO()
    out = harness.evaluate(samples(), gating, experts, pose_log=log, hypotheses=128, generator=gen)
    assert out["images"] == 8 and 1 <= out["avg_active"] <= E
    for row in out["scenes"]:
        assert row["class_acc"] == 1.0 and row["pose_acc"] == 1.0
        assert row["median_rot_deg"] < 1.0 and row["median_trans_cm"] < 3.0
    assert len(log.getvalue().strip().splitlines()) == 8
    # --expertselection / --oracleselection path: stride-0 assignment
    f = S.make_frame(300, E=E, true_expert=2)
    store["coords"] = torch.from_numpy(f["coords"]).cuda()
    store["log_gating"] = torch.log_softmax(torch.zeros(1, E, device="cuda"), dim=1)
    one = harness.localize(torch.zeros(1, 3, 480, 640, device="cuda"), gating, experts, f["focal"], hypotheses=64,
                           oracle_expert=2, generator=gen)
    assert one["expert"] == 2 and one["active_experts"] == 1
    r, t = harness.pose_errors_deg_cm(one["pose"].numpy(), f["gt_pose"])
    assert r < 1.0 and t < 3.0
