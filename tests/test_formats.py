"""On-disk formats either side of the path (esac_amd/formats.py): round trips and the reference's
conventions (room_dataset.py, expert_ensemble.py, README.md:105-120)."""
import math
import os

import numpy as np
import pytest
import torch

from esac_amd import formats as F
from esac_amd import harness
from esac_amd import synthetic as S


def test_pose_and_calibration_text_files(tmp_path):
    f = S.make_frame(0)
    p = tmp_path / "frame-000000.pose.txt"
    F.write_pose_file(p, f["gt_pose"])
    got = F.read_pose_file(p)
    assert got.dtype == torch.float32 and tuple(got.shape) == (4, 4)
    np.testing.assert_allclose(got.numpy(), f["gt_pose"].astype(np.float32), rtol=0, atol=1e-7)
    np.testing.assert_array_equal(got.numpy(), np.loadtxt(p).astype(np.float32))  # what the reference's loader reads
    c = tmp_path / "frame-000000.calibration.txt"
    F.write_calibration(c, 585.0)
    assert F.read_calibration(c) == 585.0
    assert F.read_calibration(c, image_scale=480 / 960) == 292.5  # focal scales with the image (room_dataset.py:160-164)
    (tmp_path / "bad.txt").write_text("1 2 3\n4 5 6\n")
    with pytest.raises(ValueError):
        F.read_pose_file(tmp_path / "bad.txt")


def test_init_files_and_invalid_points(tmp_path):
    f = S.make_frame(1)
    coords = torch.from_numpy(f["coords"][0]).clone()
    coords[:, 5, 7] = 0  # a hole in the depth map
    coords[:, 20:22, :] = 0
    path = tmp_path / "frame-000000.init.dat"
    F.save_init_coords(path, coords)
    back = F.load_init_coords(path)
    assert torch.equal(back, coords) and torch.equal(torch.load(path), coords)  # plain torch.save format
    off = torch.tensor([1.5, -2.0, 0.25])
    shifted = F.shift_valid_coords(back, off)
    # reference semantics (room_dataset.py:194-205)
    flat = coords.view(3, -1)
    mask = flat.abs().sum(0) == 0
    want = flat - off.unsqueeze(1).expand(flat.size())
    want[:, mask] = 0
    assert torch.equal(shifted, want.view(coords.size()))
    assert float(shifted[:, 5, 7].abs().sum()) == 0 and int(mask.sum()) == 1 + 2 * coords.size(2)
    with pytest.raises(ValueError):
        F.save_init_coords(path, torch.zeros(2, 3))
        F.load_init_coords(path)


def test_environment_grid_offsets(tmp_path):
    env = tmp_path / "env_list.txt"
    env.write_text("7scenes_chess -0.006378 -0.158068 1.608667\n7scenes_fire 0.1 0.2 0.3\nbare_scene\n\n")
    scenes, means = F.read_env_list(env)
    assert scenes == ["7scenes_chess", "7scenes_fire", "bare_scene"]
    np.testing.assert_allclose(means[0].numpy(), [-0.006378, -0.158068, 1.608667], rtol=1e-6)
    assert torch.equal(means[2], torch.zeros(3))
    n = 19  # the paper's 19Scenes environment: 5 x 5 grid of 5 m cells
    grid = math.ceil(math.sqrt(n))
    seen = set()
    for k in range(n):
        off = F.scene_offset(k, n, [0.0, 0.0, 0.0])
        row, col = math.ceil((k + 1) / grid) - 1, k % grid
        assert off.tolist() == [row * 5.0, col * 5.0, 0.0]
        seen.add((row, col))
    assert len(seen) == n
    off = F.scene_offset(7, n, [1.0, 2.0, 3.0], grid_cell_size=5.0)
    assert off.tolist() == [1.0 + 5.0, 2.0 + 10.0, 3.0]
    assert F.scene_offset(7, n, [1.0, 2.0, 3.0], normalize_mean=False).tolist() == [5.0, 10.0, 0.0]


def test_dataset_folder_listing(tmp_path):
    for split, with_init in (("training", True), ("test", False)):
        for sub in ("rgb", "calibration", "poses") + (("init",) if with_init else ()):
            d = tmp_path / "scene" / split / sub
            d.mkdir(parents=True)
            for k in (2, 0, 1):  # created out of order: matching is alphabetical
                (d / ("frame-%06d.%s" % (k, sub))).write_text("0")
    frames = F.list_frames(str(tmp_path / "scene"), "training")
    assert [os.path.basename(r["rgb"]) for r in frames] == ["frame-%06d.rgb" % k for k in range(3)]
    assert all(os.path.basename(r["init"]).endswith(".init") for r in frames)
    assert all(r["init"] is None for r in F.list_frames(str(tmp_path / "scene"), "test"))
    os.remove(tmp_path / "scene" / "test" / "poses" / "frame-000001.poses")
    with pytest.raises(ValueError):
        F.list_frames(str(tmp_path / "scene"), "test")


def test_ensemble_checkpoint_round_trip(tmp_path):
    def nets(seed):
        torch.manual_seed(seed)
        g = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Conv2d(4, 3, 1))
        es = [torch.nn.Conv2d(3, 3, 3) for _ in range(3)]
        return g, es
    g, es = nets(0)
    path = tmp_path / "esac_chess.net"
    F.save_ensemble(path, g, es)
    raw = torch.load(path)
    assert isinstance(raw, list) and len(raw) == 4 and set(raw[1]) == {"weight", "bias"}  # expert_ensemble.py:84-93
    g2, es2 = nets(1)
    assert not torch.equal(es2[2].weight, es[2].weight)
    F.load_ensemble(path, g2, es2)
    assert all(torch.equal(a, b) for a, b in zip(g.state_dict().values(), g2.state_dict().values()))
    assert all(torch.equal(e.weight, e2.weight) for e, e2 in zip(es, es2))
    with pytest.raises(ValueError):
        F.load_ensemble(path, g2, es2[:2])


def test_log_lines():
    assert F.results_log_line(1.0, 0.5, 1.25, 3.5) == "1.000000 0.500000 1.250000 3.500000\n"
    assert F.train_log_line(12, 3.25) == "12 3.250000 \n"
    assert F.strip_file_name("/data/aachen/test/rgb/query_night_nexus5x_IMG_0001.jpg") == "IMG_0001.jpg"
    assert F.strip_file_name("seq-01/frame-000123.color.png") == "frame-000123.color.png"
    line = harness.pose_file_line(F.strip_file_name("a/b/db_77.jpg"), np.eye(4))
    assert line.split()[0] == "77.jpg"


def test_files_feed_the_path(tmp_path, oracle):
    """A frame written in the dataset's formats and read back reproduces the same estimate (CPU oracle as consumer)."""
    f = S.make_frame(2)
    F.write_pose_file(tmp_path / "p.txt", f["gt_pose"])
    F.write_calibration(tmp_path / "c.txt", f["focal"])
    F.save_init_coords(tmp_path / "i.dat", torch.from_numpy(f["coords"][0]))
    coords = F.load_init_coords(tmp_path / "i.dat").numpy()[None]
    ha = S.gating_assignment(f, 32)
    a = oracle.forward(coords, ha, focal=F.read_calibration(tmp_path / "c.txt"), ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"])
    b = oracle.forward(f["coords"], ha, focal=f["focal"], ppx=f["ppx"], ppy=f["ppy"], sub_sampling=f["sub"])
    np.testing.assert_array_equal(a["pose"], b["pose"])
    r, t = harness.pose_errors_deg_cm(a["pose"], F.read_pose_file(tmp_path / "p.txt").numpy())
    assert r < 1.0 and t < 5.0
