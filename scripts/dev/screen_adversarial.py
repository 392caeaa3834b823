#!/usr/bin/env python
"""Adversarial maps for the sampling screen (host build of esac_amd/csrc/p3p_screen.hpp vs the fp64 route,
tests/native/p3p_screen_probe.cpp mode 3): planar / fronto-parallel / warped / quantised / mis-calibrated maps on which
four random cells are near-degenerate P3P configurations (double roots of the quartic, collinear or coincident samples) or
sit next to the tau boundary.  python scripts/dev/screen_adversarial.py [tries per map, default 1e7] [mode, default 3] [quantised]
(`quantised`: the second family, quantised_maps())"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from esac_amd import synthetic as S  # noqa: E402
from tests.native import build as nb  # noqa: E402


def adversarial_maps(H=60, W=80, sub=8, focal=525.0, ppx=320.0, ppy=240.0):
    """name -> float32 [3,H,W]: scene-coordinate maps that stress the P3P screen (see module docstring)."""
    xs = np.arange(W) * sub + sub // 2
    ys = np.arange(H) * sub + sub // 2
    uu, vv = np.meshgrid(xs.astype(np.float64), ys.astype(np.float64))

    def plane(depth=2.0, tilt=(0.0, 0.0), u=uu, v=vv, f=focal):
        # points of the plane z = depth + tilt . (x, y) seen through pixel (u, v) of a camera at the origin
        dx, dy = (u - ppx) / f, (v - ppy) / f
        z = depth / (1.0 - tilt[0] * dx - tilt[1] * dy)
        return np.stack([dx * z, dy * z, z]).astype(np.float32)

    rng = np.random.default_rng(77)
    maps = {}
    maps["fronto-parallel exact"] = plane()
    maps["fronto-parallel warped 1.3x0.8"] = plane(u=(uu - ppx) * 1.3 + ppx, v=(vv - ppy) * 0.8 + ppy)
    maps["tilted plane warped"] = plane(tilt=(0.4, -0.3), u=(uu - ppx) * 0.7 + ppx + 40, v=(vv - ppy) * 1.2 + ppy)
    maps["plane noise 1e-6"] = plane() + rng.normal(0, 1e-6, (3, H, W)).astype(np.float32)
    maps["plane focal 600"] = plane(f=600.0)
    maps["plane shifted 1 cell"] = np.roll(plane(tilt=(0.2, 0.1)), 1, axis=2)
    maps["plane shifted 2 cells"] = np.roll(plane(tilt=(0.2, 0.1)), 2, axis=2)
    q = plane(tilt=(0.1, 0.0)).copy()
    q[0] = np.round(q[0] * 4) / 4  # x quantised to 25 cm: many coincident / collinear samples
    maps["plane x quantised"] = q
    line = plane().copy()
    line[1] = 0.0  # y = 0, z = 2: EVERY point on one line in space -> every sample exactly collinear.  Known divergence of
    # the device's exact route from the oracle here: Horn's eigenvector alignment returns a pose with
    # an arbitrary roll about the line, which reprojects four collinear points correctly and is accepted by the reference
    # ~2e-5 of the time; the device's triad alignment yields NaN and never accepts.  Host probe only.
    maps["points on a line"] = line
    dx, dy = (uu - ppx) / focal, (vv - ppy) / focal
    d = np.stack([dx, dy, np.ones_like(dx)])
    d /= np.linalg.norm(d, axis=0)
    maps["sphere around the camera (r=2)"] = (2.0 * d).astype(np.float32)  # every scene triangle similar to its bearing triangle
    maps["sphere warped"] = np.roll((2.0 * d).astype(np.float32), 3, axis=1)
    # low acceptance (1e-4 .. 1e-2 per try): hypotheses on these walk the screened chain for thousands of tries
    maps["plane warped 3x0.33 tilted"] = plane(tilt=(0.5, 0.2), u=(uu - ppx) * 3.0 + ppx, v=(vv - ppy) * 0.33 + ppy)
    maps["plane warped 2x0.5"] = plane(u=(uu - ppx) * 2.0 + ppx, v=(vv - ppy) * 0.5 + ppy)
    nl = plane().copy()
    nl[1] = rng.normal(0, 1e-3, (H, W))
    nl[2] = 2 + rng.normal(0, 1e-3, (H, W))
    maps["points within 1 mm of a line"] = nl.astype(np.float32)  # every sample a sliver triangle
    qw = plane(u=(uu - ppx) * 1.6 + ppx, v=(vv - ppy) * 0.6 + ppy).copy()
    qw[0], qw[1] = np.round(qw[0] * 4) / 4, np.round(qw[1] * 4) / 4
    maps["plane warped, x and y quantised"] = qw  # coincident and collinear samples
    dw = np.stack([(uu - ppx) * 2.0 / focal, (vv - ppy) * 0.5 / focal, np.ones_like(uu)])
    dw /= np.linalg.norm(dw, axis=0)
    maps["sphere warped 2x0.5"] = (2.0 * dw).astype(np.float32)
    f3 = S.make_frame(3)
    c = f3["coords"][0].copy()
    maps["room, rows swapped pairwise"] = c[:, np.arange(H) ^ 1, :]
    maps["room, 1 cell right"] = np.roll(c, 1, axis=2)
    return maps


def quantised_maps():
    """A second family (round 3, after the campaign's one find came from a map with repeated scene points): predictions with
    coordinates rounded to a grid or constant over blocks of cells -- 4th cells that repeat a base point's coordinates."""
    maps = {}
    room = S.make_frame(3)["coords"][0]
    for q in (0.01, 0.05, 0.25):
        maps["room quantised %.2f m" % q] = (np.round(room / q) * q).astype(np.float32)
    for b in (2, 4):
        blk = room[:, ::b, ::b]
        maps["room piecewise constant %dx%d" % (b, b)] = np.repeat(np.repeat(blk, b, axis=1), b, axis=2)[:, :60, :80].copy()
    clean = S.make_frame(5, noise=0.0, outlier_frac=0.0)["coords"][0]
    maps["clean room quantised 0.05 m"] = (np.round(clean / 0.05) * 0.05).astype(np.float32)
    blk = clean[:, ::2, ::2]
    maps["clean room piecewise constant 2x2"] = np.repeat(np.repeat(blk, 2, axis=1), 2, axis=2)[:, :60, :80].copy()
    am = adversarial_maps()
    maps["sphere warped quantised 0.125"] = (np.round(am["sphere warped"] * 8) / 8).astype(np.float32)
    maps["plane warped 2x0.5 quantised xyz 0.25"] = (np.round(am["plane warped 2x0.5"] * 4) / 4).astype(np.float32)
    return maps


def run(lib, coords, n, seed, sub=8, focal=525.0, ppx=320.0, ppy=240.0, tau=10.0, margins=(0.5, 1.0, 2.0, 3.0), mode=3):
    m = np.asarray(margins, np.float32)
    out = np.zeros(40)
    c = np.ascontiguousarray(coords, np.float32)
    _, H, W = c.shape
    lib.probe_screen(c.ctypes.data_as(C.c_void_p), H, W, sub, 0, 0, C.c_float(focal), C.c_float(ppx), C.c_float(ppy), C.c_float(tau),
                     C.c_uint64(seed), C.c_longlong(int(n)), m.ctypes.data_as(C.c_void_p), len(m), mode, out.ctypes.data_as(C.c_void_p))
    return out


if __name__ == "__main__":
    n = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    lib = C.CDLL(nb.build_screen_probe())
    tot = rej = 0
    family = quantised_maps() if len(sys.argv) > 3 and sys.argv[3] == "quantised" else adversarial_maps()
    for k, (name, coords) in enumerate(family.items()):
        out = run(lib, coords, n, (900 if family is not None and len(sys.argv) > 3 else 500) + k, mode=mode)
        t, acc = out[0], out[1]
        tot += acc
        rej += out[15]
        print("      guards fired (1 A, 2 R2, 3 D2/E2, 4 b0, 5 root, 6 b1, 7 b0 rel, 8 v): " + " ".join("%.4f%%" % (100 * v / t) for v in out[30:38]))
        print("%-34s accepted %9d (%.2e)  maybe@3px %.5f  delicate %.3f%%  false rejects @0.5/1/2/3 px: %d %d %d %d  max screen err of accepted %.3f" % (
            name, acc, acc / t, out[7] / t, 100 * out[2] / t, out[12], out[13], out[14], out[15], out[3]))
    print("total fp64-accepted tries %d, false rejects at the kernel's margin %d" % (tot, rej))
