"""Seeded synthetic "3D-print tile" images (no dataset exists in the reference snapshot).

HR = clip(0.5 + 0.22 sin(2 pi y / p_y + phi) + 0.08 sin(2 pi x / p_x) + 0.05 box3(N(0,1)) + sparse dark
blobs, 0, 1), RGB-tinted per tile; LR = s x s box mean of HR.
"""
import numpy as np


def _box3(a):
    p = np.pad(a, 1, mode="edge")
    return sum(p[i:i + a.shape[0], j:j + a.shape[1]] for i in range(3) for j in range(3)) / 9.0


def hr_tile(rng, h, w):
    y = np.arange(h, dtype=np.float64)[:, None]
    x = np.arange(w, dtype=np.float64)[None, :]
    py, px = rng.uniform(6, 14), rng.uniform(20, 60)
    base = 0.5 + 0.22 * np.sin(2 * np.pi * y / py + rng.uniform(0, 2 * np.pi)) + 0.08 * np.sin(2 * np.pi * x / px)
    base = base + 0.05 * _box3(rng.standard_normal((h, w)))
    for _ in range(int(rng.integers(2, 6))):          # sparse dark blobs (defects)
        cy, cx, r = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(2, max(3, min(h, w) / 16))
        base = base - 0.35 * np.exp(-((y - cy) ** 2 + (x - cx) ** 2) / (2 * r * r))
    tint = rng.uniform(0.85, 1.0, size=3)
    return np.clip(base[:, :, None] * tint[None, None, :], 0.0, 1.0).astype(np.float32)


def box_down(hr, s):
    h, w, c = hr.shape
    return hr.reshape(h // s, s, w // s, s, c).mean(axis=(1, 3)).astype(np.float32)


def make_pairs(n, lr_h, lr_w, scale, seed):
    """-> (LR [n,lr_h,lr_w,3], HR [n,lr_h*scale,lr_w*scale,3]) float32 in [0,1]."""
    rng = np.random.default_rng(seed)
    hr = np.stack([hr_tile(rng, lr_h * scale, lr_w * scale) for _ in range(n)])
    lr = np.stack([box_down(t, scale) for t in hr])
    return lr, hr
