"""Deterministic synthetic depth generators shared by the tests, __graft_entry__.smoke() and bench.py.

synth_depth: piecewise-planar "room" seen through a pinhole: 4-8 random planes facing the camera,
depth = nearest intersection per pixel, raw units of 1 mm (x`scale`), Gaussian noise sigma = 1.425e-6 * z^2
(the reference's depth_sigma_coeff), 5-15 % dropout as blobs and along plane intersections, clipped to uint16.
tum_stream: the reference's single TUM frame plus seeded variants (per-pixel integer noise on valid pixels and a
small circular shift) standing in for "TUM fr3 depth stream" (BASELINE.json configs[1]).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden" / "data"
TUM_PNG = GOLDEN / "tum" / "1341848230.910894.png"
TUM_K = GOLDEN / "configs" / "TUM_fr3_long_val.K"
ICL_PNG = GOLDEN / "icl_nuim" / "0.png"
ICL_K = GOLDEN / "configs" / "ICL_living_room.K"
ICL_INI = GOLDEN / "configs" / "ICL_living_room.ini"
TUM_INI = GOLDEN / "configs" / "TUM_fr3_long_val.ini"


def read_K(path) -> np.ndarray:
    return np.array(open(path).read().split(), dtype=np.float32).reshape(3, 3)


def intrinsics_for(h: int, w: int):
    """fx = fy = 0.835 W, principal point at the image centre."""
    return float(np.float32(0.835 * w)), float(np.float32(0.835 * w)), float(np.float32(w / 2 - 0.5)), float(np.float32(h / 2 - 0.5))


def load_png_u16(path) -> np.ndarray:
    from PIL import Image
    return np.array(Image.open(path)).astype(np.uint16)


def synth_depth(h: int, w: int, seed: int = 0, noise: bool = True, dropout: bool = True, scale: float = 1.0) -> np.ndarray:
    rng = np.random.default_rng(0xD1B54A32D192ED03 ^ int(seed))
    fx, fy, cx, cy = intrinsics_for(h, w)
    u = (np.arange(w, dtype=np.float64) - cx) / fx
    v = (np.arange(h, dtype=np.float64) - cy) / fy
    dx, dy = np.meshgrid(u, v)
    nplanes = int(rng.integers(4, 9))
    zs = []
    for _ in range(nplanes):
        # normal within 75 degrees of the optical axis, facing the camera (n_z < 0)
        tilt = np.deg2rad(rng.uniform(0.0, 75.0))
        az = rng.uniform(0.0, 2 * np.pi)
        n = np.array([np.sin(tilt) * np.cos(az), np.sin(tilt) * np.sin(az), -np.cos(tilt)])
        d = rng.uniform(500.0, 6000.0)  # mm
        ndir = n[0] * dx + n[1] * dy + n[2]
        z = np.where(ndir < -1e-6, d / np.maximum(-ndir, 1e-6), np.inf)
        zs.append(z)
    zs = np.stack(zs)
    order = np.sort(zs, axis=0)
    z = order[0].copy()
    z[~np.isfinite(z)] = 0.0
    if noise:
        z = z + rng.standard_normal(z.shape) * (1.425e-6 * z * z)
    z = z * scale
    if dropout:
        drop = np.zeros((h, w), dtype=bool)
        target = rng.uniform(0.05, 0.15)
        # plane intersections: the two nearest surfaces closer than 1 %
        gap = (order[1] - order[0]) / np.maximum(order[0], 1.0)
        drop |= gap < 0.01
        yy, xx = np.mgrid[0:h, 0:w]
        tries = 0
        while drop.mean() < target and tries < 200:
            r = rng.uniform(0.01, 0.06) * w
            c0, r0 = rng.uniform(0, w), rng.uniform(0, h)
            drop |= (xx - c0) ** 2 + (yy - r0) ** 2 < r * r
            tries += 1
        z[drop] = 0.0
    return np.clip(np.rint(z), 0, 65535).astype(np.uint16)


def quilt_depth(h: int, w: int, seed: int = 0, P: int = 10, kmax: int = 3, ntilt: int = 6, jitter: float = 0.0, drop: float = 0.05) -> np.ndarray:
    """A quilt of small planar patches (1..kmax cells of P pixels on a side) whose tilts come from a handful of directions
    and whose depths partly coincide: many histogram bins hold the same number of cells, most grows are a few cells next
    to other small grows, and some span patches.  The worst case for the sweep rounds of the region kernel (many events
    per histogram level, grows that touch, bins that lose cells in the middle of a sweep): 2 600 - 22 500 grows and
    1 100 - 7 700 plane segments per frame where the piecewise-planar rooms of synth_depth have a few hundred."""
    rng = np.random.default_rng(0x9E3779B97F4A7C15 ^ int(seed))
    fx, fy, cx, cy = intrinsics_for(h, w)
    nh, nv = w // P, h // P
    tilts = []
    for _ in range(ntilt):
        t, a = np.deg2rad(rng.uniform(5, 55)), rng.uniform(0, 2 * np.pi)
        tilts.append((np.tan(t) * np.cos(a), np.tan(t) * np.sin(a)))
    z = np.zeros((h, w))
    yy, xx = np.mgrid[0:h, 0:w]
    r = 0
    while r < nv:
        kh = int(rng.integers(1, kmax + 1))
        c = 0
        while c < nh:
            kw = int(rng.integers(1, kmax + 1))
            gx, gy = tilts[int(rng.integers(0, ntilt))]
            z0 = rng.uniform(1500, 4000) if rng.random() < 0.5 else 2500 + 40 * ((r + c) % 5)
            ys, xs = slice(r * P, min((r + kh) * P, h)), slice(c * P, min((c + kw) * P, w))
            z[ys, xs] = z0 * (1 + gx * (xx[ys, xs] - c * P) / fx + gy * (yy[ys, xs] - r * P) / fy)
            c += kw
        r += kh
    if jitter:
        z += rng.standard_normal(z.shape) * jitter
    dropm = rng.random((nv + 1, nh + 1)) < drop
    z[np.repeat(np.repeat(dropm, P, 0), P, 1)[:h, :w]] = 0
    return np.clip(np.rint(z), 0, 65535).astype(np.uint16)


def synth_batch(batch: int, h: int, w: int, seed0: int = 0, **kw) -> np.ndarray:
    return np.stack([synth_depth(h, w, seed=seed0 + i, **kw) for i in range(batch)])


def tum_stream(batch: int, seed0: int = 0) -> np.ndarray:
    """Frame 0 is the reference's TUM frame unmodified; frame k>0 adds integer noise in [-k', k'] (k' = 1 + k % 4)
    to valid pixels and rolls the image by (k % 5 - 2, k % 7 - 3) pixels; seed = frame index."""
    base = load_png_u16(TUM_PNG)
    out = np.empty((batch,) + base.shape, dtype=np.uint16)
    for k in range(batch):
        if k == 0 and seed0 == 0:
            out[k] = base
            continue
        rng = np.random.default_rng(seed0 + k)
        amp = 1 + (k % 4)
        img = base.astype(np.int32)
        noise = rng.integers(-amp, amp + 1, size=img.shape)
        img = np.where(img > 0, np.clip(img + noise, 1, 65535), 0)
        img = np.roll(img, (k % 5 - 2, k % 7 - 3), axis=(0, 1))
        out[k] = img.astype(np.uint16)
    return out
