"""Caller-side glue of the hot path: the evaluation loop of code/test_esac.py:135-289 with every tensor
device-resident.

What changes against the reference loop (and why it matters on MI355X):
  * `prediction = prediction.cpu()` (test_esac.py:187) is gone: the coordinate maps stay in HBM and go straight
    into `esac.forward`; `gating_probs.cpu()` (test_esac.py:164) is gone: `torch.multinomial` / `torch.histc`
    run on the device; only the per-expert activity mask (E booleans) and the result record cross PCIe.
  * experts are evaluated only where the gating drew at least one hypothesis (test_esac.py:178-185), as before.
  * the pose error (test_esac.py:209-217) and the quaternion pose-file line (test_esac.py:230-247) no longer
    need cv2: Rodrigues is a few lines of numpy (cv2 / skimage / torchvision are not installed here).

The expert and gating networks are whatever `nn.Module`s the caller passes (the reference's `Expert` /
`Gating` FCNs stay in PyTorch-ROCm, north_star); the tests drive the loop with synthetic experts that return
the ray-cast maps of esac_amd/synthetic.py, since no datasets or trained weights exist offline.
"""
import math
import time

import numpy as np
import torch

from . import api


def rodrigues_vector(R):
    """Rotation matrix -> axis-angle vector (what cv2.Rodrigues(R)[0] returns), numpy only."""
    R = np.asarray(R, np.float64)
    sk = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.linalg.norm(sk)
    c = (np.trace(R) - 1.0) / 2.0
    angle = math.atan2(s, c)
    if s < 1e-12:
        if c > 0:
            return np.zeros(3)
        # angle ~ pi: axis from the diagonal of (R + I) / 2
        d = np.clip((np.diag(R) + 1.0) / 2.0, 0.0, None)
        axis = np.sqrt(d)
        if R[0, 1] < 0:
            axis[1] = -axis[1]
        if R[0, 2] < 0:
            axis[2] = -axis[2]
        return axis / max(np.linalg.norm(axis), 1e-300) * angle
    return sk / s * angle


def pose_errors_deg_cm(out_pose, gt_pose):
    """(rotation error in degrees, translation error in cm) exactly as test_esac.py:209-217 prints them."""
    out_pose = np.asarray(out_pose, np.float64)
    gt_pose = np.asarray(gt_pose, np.float64)
    t_err = float(np.linalg.norm(gt_pose[0:3, 3] - out_pose[0:3, 3]))
    r = rodrigues_vector(out_pose[0:3, 0:3] @ gt_pose[0:3, 0:3].T)
    return float(np.linalg.norm(r) * 180.0 / math.pi), t_err * 100.0


def pose_file_line(name, out_pose):
    """One line of poses_esac_<session>.txt: name qw qx qy qz tx ty tz of the INVERTED pose (test_esac.py:230-247)."""
    inv = np.linalg.inv(np.asarray(out_pose, np.float64))
    t = inv[0:3, 3]
    rot = rodrigues_vector(inv[0:3, 0:3])
    angle = float(np.linalg.norm(rot))
    axis = rot / angle if angle > 0 else np.array([1.0, 0.0, 0.0])
    q_w = math.cos(angle * 0.5)
    q_xyz = math.sin(angle * 0.5) * axis
    return "%s %f %f %f %f %f %f %f\n" % (name, q_w, q_xyz[0], q_xyz[1], q_xyz[2], float(t[0]), float(t[1]), float(t[2]))


@torch.no_grad()
def localize(image, gating, experts, focal_length, hypotheses=256, threshold=10.0, inlier_alpha=100.0,
             inlier_beta=0.5, max_reprojection=100.0, subsample=8, expert_selection=False, oracle_expert=None,
             generator=None):
    """One iteration of the reference test loop (test_esac.py:145-207) for `image` [1,3,H,W] on the GPU.

    gating(image) -> log-probabilities [1,E]; experts[e](image) -> scene coordinates [1,3,H/s,W/s].
    Returns dict(pose [4,4] float32 cpu, expert, active_experts, gating_probs (device), time_s, prediction, hyp_assignment
    -- the two device tensors handed to esac.forward, so a caller can replay the call elsewhere)."""
    dev = image.device
    E = len(experts)
    pp_x = float(image.size(3) / 2)
    pp_y = float(image.size(2) / 2)
    pred_w = math.ceil(image.size(3) / subsample)
    pred_h = math.ceil(image.size(2) / subsample)
    prediction = torch.zeros((E, 3, pred_h, pred_w), device=dev)
    start = time.time()
    gating_probs = torch.exp(gating(image))[0]  # stays on the device (reference: .cpu())
    if oracle_expert is not None:
        gating_probs = torch.zeros_like(gating_probs)
        gating_probs[int(oracle_expert)] = 1
    if expert_selection or oracle_expert is not None:
        expert = torch.multinomial(gating_probs, 1, replacement=True, generator=generator)
        e_hyps = expert.expand((hypotheses,))  # stride-0 view, as in the reference
    else:
        e_hyps = torch.multinomial(gating_probs, hypotheses, replacement=True, generator=generator)
    e_hist = torch.histc(e_hyps.float(), bins=E, min=0, max=E - 1)
    active = (e_hist > 0).cpu().tolist()  # E booleans: the only device->host traffic before the call
    for e, on in enumerate(active):
        if on:
            prediction[e] = experts[e](image)[0]
    out_pose = torch.zeros(4, 4)
    winning_expert = api.forward(prediction, e_hyps, out_pose, 0, 0, float(focal_length), pp_x, pp_y, threshold,
                                 inlier_alpha, inlier_beta, max_reprojection, subsample)
    return dict(pose=out_pose, expert=winning_expert, active_experts=int(sum(active)), gating_probs=gating_probs,
                time_s=time.time() - start, prediction=prediction, hyp_assignment=e_hyps)


def evaluate(samples, gating, experts, trans_threshold_cm=5.0, rot_threshold_deg=5.0, pose_log=None, **kw):
    """The statistics block of test_esac.py:249-289 over `samples` = iterable of
    (name, image, focal_length, gt_pose [4,4], gt_expert)."""
    E = len(experts)
    scenes_r, scenes_t, scenes_c = [[] for _ in range(E)], [[] for _ in range(E)], [[] for _ in range(E)]
    avg_active = max_active = avg_time = n = 0
    for name, image, focal, gt_pose, gt_expert in samples:
        out = localize(image, gating, experts, focal, **kw)
        r_err, t_err = pose_errors_deg_cm(out["pose"].numpy(), np.asarray(gt_pose))
        scenes_r[gt_expert].append(r_err)
        scenes_t[gt_expert].append(t_err)
        scenes_c[gt_expert].append(int(gt_expert) == out["expert"])
        avg_active += out["active_experts"]
        max_active = max(max_active, out["active_experts"])
        avg_time += out["time_s"]
        n += 1
        if pose_log is not None:
            pose_log.write(pose_file_line(name, out["pose"].numpy()))

    def median(values):
        if len(values) == 0:
            return 0
        values = sorted(values)
        return values[int(len(values) / 2)]

    rows = []
    for s in range(E):
        class_acc = sum(scenes_c[s]) / max(len(scenes_c[s]), 1)
        ok = [(t < trans_threshold_cm and r < rot_threshold_deg) for t, r in zip(scenes_t[s], scenes_r[s])]
        rows.append(dict(scene=s, class_acc=class_acc, pose_acc=sum(ok) / max(len(ok), 1),
                         median_rot_deg=median(scenes_r[s]), median_trans_cm=median(scenes_t[s])))
    return dict(scenes=rows, avg_active=avg_active / max(n, 1), max_active=max_active, avg_time_s=avg_time / max(n, 1),
                images=n)


# ---------------------------------------------------------------- training glue (train_esac.py:104-200)
def random_shift(image, max_shift, rng=None):
    """Zero-pad shift augmentation of util.py:4-11: returns (padX, padY, shifted image)."""
    import random
    r = rng if rng is not None else random
    pad_x = r.randint(-int(max_shift), int(max_shift))
    pad_y = r.randint(-int(max_shift), int(max_shift))
    return pad_x, pad_y, torch.nn.functional.pad(image, (pad_x, -pad_x, pad_y, -pad_y))


def clamp_probs(probs, n):
    """Zero all but the n largest entries in place (util.py:38-47); one topk instead of a Python loop over E."""
    if n < 0 or n >= probs.numel():
        return
    keep = torch.zeros_like(probs, dtype=torch.bool)
    if n > 0:
        keep[torch.topk(probs, n).indices] = True
    probs.masked_fill_(~keep, 0)


def train_step(image, gt_pose, gating, experts, focal_length, hypotheses=256, threshold=10.0, inlier_alpha=100.0,
               inlier_beta=0.5, max_reprojection=100.0, subsample=8, weight_rot=1.0, weight_trans=100.0, loss_cut=100.0,
               max_experts=-1, expert_selection=False, shift=None, generator=None):
    """One iteration of the end-to-end training loop (train_esac.py:104-192) up to and including
    `torch.autograd.backward`; the optimiser step stays with the caller (`ensemble.update`, train_esac.py:195).

    Differences from the reference loop, all of them removals of host round trips:
      * `prediction.cpu()` (train_esac.py:152) and `prediction_gradients.cuda()` (:185) are gone -- the coordinate
        tensor and its gradient container stay in HBM and `esac.backward` accumulates into the latter in place;
      * `torch.exp(gating_log_probs).cpu()` (:129) is gone -- clamp / multinomial / histc run on the device, only the
        E activity flags and the loss value reach the host.
    gating(image) -> log-probabilities [1,E] (with grad); experts[e](image) -> [1,3,H/s,W/s] (with grad).
    Returns dict(loss, e_hyps, e_hist, prediction, prediction_gradients, gating_log_probs, pad)."""
    dev = image.device
    E = len(experts)
    pp_x = float(image.size(3) / 2)
    pp_y = float(image.size(2) / 2)
    pred_w = math.ceil(image.size(3) / subsample)
    pred_h = math.ceil(image.size(2) / subsample)
    if shift is None:
        pad_x, pad_y, image = random_shift(image, subsample / 2)
    else:
        pad_x, pad_y = int(shift[0]), int(shift[1])
        image = torch.nn.functional.pad(image, (pad_x, -pad_x, pad_y, -pad_y))
    gating_log_probs = gating(image)
    with torch.no_grad():
        gating_probs = torch.exp(gating_log_probs)[0].clone()
        clamp_probs(gating_probs, max_experts)
        if expert_selection:
            expert = torch.multinomial(gating_probs, 1, replacement=True, generator=generator)
            e_hyps = expert.expand((hypotheses,))
        else:
            e_hyps = torch.multinomial(gating_probs, hypotheses, replacement=True, generator=generator)
        e_hist = torch.histc(e_hyps.float(), bins=E, min=0, max=E - 1)
        active = (e_hist > 0).cpu().tolist()
    outputs = [experts[e](image)[0] if on else torch.zeros((3, pred_h, pred_w), device=dev) for e, on in enumerate(active)]
    prediction = torch.stack(outputs)  # [E,3,h,w]; rows of inactive experts are zeros and never read
    prediction_gradients = torch.zeros_like(prediction)
    loss = api.backward(prediction.detach(), prediction_gradients, e_hyps, torch.as_tensor(gt_pose, dtype=torch.float32).cpu(),
                        weight_rot, weight_trans, loss_cut, pad_x, pad_y, float(focal_length), pp_x, pp_y, threshold,
                        inlier_alpha, inlier_beta, max_reprojection, subsample)
    # gating gradients: REINFORCE-style, loss per drawn hypothesis (train_esac.py:171-177)
    if expert_selection:
        gating_grads = torch.zeros_like(gating_log_probs)
        gating_grads[0, int(expert)] = loss
    else:
        gating_grads = (loss * e_hist).unsqueeze(0).to(gating_log_probs.dtype)
    tensors, grads = [], []
    if prediction.requires_grad:
        tensors.append(prediction)
        grads.append(prediction_gradients)
    if gating_log_probs.requires_grad:
        tensors.append(gating_log_probs)
        grads.append(gating_grads)
    if tensors:
        torch.autograd.backward(tensors, grads)
    return dict(loss=loss, e_hyps=e_hyps, e_hist=e_hist, prediction=prediction, prediction_gradients=prediction_gradients,
                gating_log_probs=gating_log_probs, pad=(pad_x, pad_y))
