"""Synthetic frames for the hot path (no datasets / pretrained experts exist offline).

Recipe = SURVEY.md section 8(d): 640x480 camera, f=525, pp=(320,240), output
sub-sampling 8 -> 60x80 grid of pixel centres (8x+4, 8y+4) (esac_util.h:64-66);
an inward-facing 4x3x4 m box room around the 7-Scenes `chess` centre
(environments/7scenes/env_list.txt:1), ray-cast per grid cell for exact scene
coordinates, N(0, 2 cm) noise on every point, 30 % uniform outliers.  Experts
other than the true one predict N(own centre, 1 m) garbage unrelated to the
image.  Frame k is seeded with numpy default_rng(1000+k).
"""
import math

import numpy as np

CHESS_CENTRE = np.array([-0.006378, -0.158068, 1.608667])
ROOM_HALF = np.array([2.0, 1.5, 2.0])


def _rodrigues(axis, angle):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + math.sin(angle) * K + (1 - math.cos(angle)) * (K @ K)


def make_frame(k=0, E=1, true_expert=0, H=60, W=80, sub=8, focal=525.0, ppx=320.0, ppy=240.0,
               noise=0.02, outlier_frac=0.3, shift=(0, 0), grid_spacing=5.0):
    """Returns dict(coords[E,3,H,W] float32, gt_pose[4,4] float64 camera->scene, ...)."""
    rng = np.random.default_rng(1000 + k)
    # expert centres: rooms laid out on a 5 m grid (room_dataset.py:170-187)
    centres = np.stack([CHESS_CENTRE + np.array([grid_spacing * (e % 5), 0.0, grid_spacing * (e // 5)])
                        for e in range(E)])
    c0 = centres[true_expert]
    axis = rng.normal(size=3)
    angle = rng.uniform(0.0, math.radians(30.0))
    R_c2s = _rodrigues(axis, angle)
    cam = c0 + rng.uniform(-1.0, 1.0, size=3)
    xs = np.arange(W) * sub + sub // 2 - shift[0]
    ys = np.arange(H) * sub + sub // 2 - shift[1]
    uu, vv = np.meshgrid(xs, ys)  # [H,W]
    d_cam = np.stack([(uu - ppx) / focal, (vv - ppy) / focal, np.ones_like(uu, np.float64)], -1)
    d_w = d_cam @ R_c2s.T  # [H,W,3]
    lo, hi = c0 - ROOM_HALF, c0 + ROOM_HALF
    with np.errstate(divide="ignore", invalid="ignore"):
        t_lo = (lo - cam) / d_w
        t_hi = (hi - cam) / d_w
    t_exit = np.minimum(np.where(d_w > 0, t_hi, np.inf), np.where(d_w < 0, t_lo, np.inf)).min(-1)
    pts = cam + d_w * t_exit[..., None]
    pts = pts + rng.normal(0.0, noise, size=pts.shape)
    out_mask = rng.uniform(size=(H, W)) < outlier_frac
    outl = rng.uniform(lo, hi, size=(H, W, 3))
    pts = np.where(out_mask[..., None], outl, pts)
    coords = np.zeros((E, 3, H, W), np.float32)
    for e in range(E):
        if e == true_expert:
            coords[e] = np.transpose(pts, (2, 0, 1)).astype(np.float32)
        else:
            g = centres[e] + rng.normal(0.0, 1.0, size=(H, W, 3))
            coords[e] = np.transpose(g, (2, 0, 1)).astype(np.float32)
    gt = np.eye(4)
    gt[:3, :3] = R_c2s
    gt[:3, 3] = cam
    return dict(coords=coords, gt_pose=gt, focal=float(focal), ppx=float(ppx), ppy=float(ppy), sub=int(sub),
                shift=tuple(shift), true_expert=int(true_expert), outlier_mask=out_mask, rng=rng)


def gating_assignment(frame, N, mode="single", rng=None):
    """hypAssignment as test_esac.py:171-175 would draw it.

    single: all hypotheses on the true expert.  gating: softmax(6 for the true
    expert, N(0,1) others) then multinomial with replacement.  dirichlet:
    Dirichlet(0.3) over experts (config 5)."""
    E = frame["coords"].shape[0]
    rng = rng if rng is not None else frame["rng"]
    if mode == "single" or E == 1:
        return np.full(N, frame["true_expert"], np.int64)
    if mode == "gating":
        logits = rng.normal(size=E)
        logits[frame["true_expert"]] = 6.0
        p = np.exp(logits - logits.max())
        p /= p.sum()
    elif mode == "dirichlet":
        p = rng.dirichlet(np.full(E, 0.3))
    else:
        raise ValueError(mode)
    return rng.choice(E, size=N, replace=True, p=p).astype(np.int64)


def pose_errors(out_pose, gt_pose):
    """(rot_err_rad, trans_err_m) as test_esac.py:209-217 (which prints degrees / cm)."""
    out_pose = np.asarray(out_pose, np.float64)
    gt_pose = np.asarray(gt_pose, np.float64)
    t_err = float(np.linalg.norm(gt_pose[:3, 3] - out_pose[:3, 3]))
    Rr = out_pose[:3, :3] @ gt_pose[:3, :3].T
    # |Rodrigues(Rr)| as cv2.Rodrigues gives it, but via atan2(sin, cos): acos alone loses half the
    # digits near 0 (float32 poses would read ~3e-4 rad for identical rotations)
    sk = 0.5 * np.array([Rr[2, 1] - Rr[1, 2], Rr[0, 2] - Rr[2, 0], Rr[1, 0] - Rr[0, 1]])
    return float(math.atan2(np.linalg.norm(sk), (np.trace(Rr) - 1.0) / 2.0)), t_err
