"""On-disk formats either side of the hot path, so that real datasets / checkpoints of the
reference can be fed to `esac.forward` / `esac.backward` and their outputs written back in the reference's formats.

  pose files         text, 4x4 camera pose                      README.md:116, room_dataset.py:167-168
  calibration files  text, one number: focal length in px       README.md:118, room_dataset.py:163-164
  init files         torch.save'd float tensor [3,H/8,W/8], all-zero points = invalid   README.md:120, room_dataset.py:190-205
  dataset folders    rgb/ calibration/ poses/ init/, matched by alphabetical order       README.md:105-112, room_dataset.py:50-89
  environment grid   scene k sits in cell (row, col) of a ceil(sqrt(n)) grid of 5 m cells room_dataset.py:170-187
  ensemble file      torch.save([gating.state_dict(), expert_0.state_dict(), ...])       expert_ensemble.py:84-112
  result logs        "<class_acc> <pose_acc> <median_rot> <median_trans>" per scene      test_esac.py:280
                     pose log lines: esac_amd/harness.py:pose_file_line                   test_esac.py:230-247
Image decoding is not here: the reference uses scikit-image, which this image does not ship; the CNNs and their
inputs stay with the caller (north_star).
"""
import math
import os

import numpy as np
import torch


# ---------------------------------------------------------------- single files
def read_pose_file(path):
    """4x4 camera pose as float32 tensor (room_dataset.py:167-168)."""
    pose = np.loadtxt(path)
    if pose.shape != (4, 4):
        raise ValueError("%s: expected a 4x4 matrix, found shape %s" % (path, pose.shape))
    return torch.from_numpy(pose).float()


def write_pose_file(path, pose):
    np.savetxt(path, np.asarray(pose, np.float64).reshape(4, 4))


def read_calibration(path, image_scale=1.0):
    """Focal length in px, scaled like the image (room_dataset.py:160-164)."""
    return float(np.loadtxt(path)) * float(image_scale)


def write_calibration(path, focal_length):
    with open(path, "w") as f:
        f.write("%f\n" % float(focal_length))


def load_init_coords(path):
    """Ground-truth scene coordinates [3,h,w] float32 (room_dataset.py:192); zeros mark invalid points."""
    t = torch.load(path, map_location="cpu")
    if t.dim() != 3 or t.size(0) != 3:
        raise ValueError("%s: expected a [3,h,w] tensor, found %s" % (path, tuple(t.shape)))
    return t.float()


def save_init_coords(path, coords):
    torch.save(torch.as_tensor(coords).float().cpu(), path)


def shift_valid_coords(coords, offset):
    """coords - offset for every valid point, invalid (all-zero) points stay zero (room_dataset.py:194-205)."""
    size = coords.size()
    flat = coords.reshape(3, -1)
    invalid = flat.abs().sum(0) == 0
    out = flat - torch.as_tensor(offset, dtype=flat.dtype).reshape(3, 1)
    out[:, invalid] = 0
    return out.reshape(size)


# ---------------------------------------------------------------- environments
def scene_offset(scene_idx, n_scenes, mean, grid_cell_size=5.0, normalize_mean=True):
    """Translation subtracted from a scene's poses and coordinates to place it in the environment
    (room_dataset.py:170-187): its mean (optional) plus its cell in a ceil(sqrt(n))-wide grid."""
    offset = torch.as_tensor(mean, dtype=torch.float32).clone()
    if not normalize_mean:
        offset.fill_(0)
    grid_size = math.ceil(math.sqrt(n_scenes))
    row = math.ceil((scene_idx + 1) / grid_size) - 1
    col = scene_idx % grid_size
    offset[0] += row * grid_cell_size
    offset[1] += col * grid_cell_size
    return offset


def read_env_list(path):
    """env_list.txt: one scene per line, `<folder> [mean_x mean_y mean_z]` (room_dataset.py:30-47)."""
    scenes, means = [], []
    with open(path) as f:
        for line in f:
            parts = line.split()
            if not parts:
                continue
            scenes.append(parts[0])
            means.append([float(v) for v in parts[1:4]] if len(parts) >= 4 else [0.0, 0.0, 0.0])
    return scenes, torch.tensor(means, dtype=torch.float32)


def list_frames(scene_dir, split="test", with_init=None):
    """Per-frame file tuples of one scene folder, matched by alphabetical order (README.md:112):
    dict(rgb, calibration, pose[, init]).  `with_init` defaults to split == 'training'."""
    root = os.path.join(scene_dir, split)
    need_init = (split == "training") if with_init is None else with_init

    def sorted_files(sub):
        d = os.path.join(root, sub)
        return [os.path.join(d, f) for f in sorted(os.listdir(d))]
    rgb, cal, pose = sorted_files("rgb"), sorted_files("calibration"), sorted_files("poses")
    init = sorted_files("init") if need_init else [None] * len(rgb)
    if not (len(rgb) == len(cal) == len(pose) == len(init)):
        raise ValueError("%s: rgb/calibration/poses/init hold %d/%d/%d/%d files" % (root, len(rgb), len(cal), len(pose), len(init)))
    return [dict(rgb=r, calibration=c, pose=p, init=i) for r, c, p, i in zip(rgb, cal, pose, init)]


# ---------------------------------------------------------------- checkpoints, logs
def save_ensemble(path, gating, experts):
    """One file = [gating state_dict, expert state_dicts...] (expert_ensemble.py:84-93)."""
    torch.save([gating.state_dict()] + [e.state_dict() for e in experts], path)


def load_ensemble(path, gating, experts, map_location=None):
    """Inverse of save_ensemble (expert_ensemble.py:105-112)."""
    state = torch.load(path, map_location=map_location)
    if len(state) != len(experts) + 1:
        raise ValueError("%s holds %d networks, the ensemble has %d" % (path, len(state), len(experts) + 1))
    gating.load_state_dict(state[0])
    for i, e in enumerate(experts):
        e.load_state_dict(state[i + 1])


def results_log_line(class_acc, pose_acc, median_rot_deg, median_trans_cm):
    """One scene of results_esac_<session>.txt (test_esac.py:280)."""
    return "%f %f %f %f\n" % (class_acc, pose_acc, median_rot_deg, median_trans_cm)


def train_log_line(iteration, loss):
    """One iteration of log_esac_<session>.txt (train_esac.py:198)."""
    return "%d %f \n" % (iteration, loss)


def strip_file_name(f):
    """Image name as written to the pose log (util.py:49-60): no path, no Aachen prefixes."""
    f = f.split("/")[-1]
    for ign in ("db_", "query_day_milestone_", "query_day_nexus4_", "query_day_nexus5x_", "query_night_nexus5x_"):
        if f.startswith(ign):
            f = f[len(ign):]
    return f
