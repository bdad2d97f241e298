"""Synthetic video used by the tests, the golden-fixture generator and bench.py.

There is no dataset on any box (no network), so every input is regenerated from a seed:
a smooth random field translated by a few pixels per frame with edge replication plus a
little sensor noise.  numpy only, so the CPU oracle, the
reference-driven fixture generator and the GPU path all see bit-identical frames.
"""
from __future__ import annotations

import numpy as np


def _smooth_field(g: np.random.Generator, c: int, h: int, w: int) -> np.ndarray:
    """Sum of bilinearly upsampled coarse noise octaves, normalised to [0, 1]."""
    out = np.zeros((c, h, w), np.float32)
    amp = 1.0
    for cells in (4, 8, 16, 32):
        gh, gw = max(2, h // cells + 2), max(2, w // cells + 2)
        coarse = g.random((c, gh, gw), dtype=np.float32)
        ys = np.linspace(0, gh - 1.001, h, dtype=np.float32)
        xs = np.linspace(0, gw - 1.001, w, dtype=np.float32)
        y0, x0 = ys.astype(np.int64), xs.astype(np.int64)
        fy, fx = (ys - y0)[None, :, None], (xs - x0)[None, None, :]
        a = coarse[:, y0][:, :, x0]
        b = coarse[:, y0][:, :, x0 + 1]
        cc = coarse[:, y0 + 1][:, :, x0]
        d = coarse[:, y0 + 1][:, :, x0 + 1]
        out += amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (cc * (1 - fx) + d * fx) * fy)
        amp *= 0.5
    out -= out.min()
    out /= max(float(out.max()), 1e-6)
    return out


def frames(seed: int, n_frames: int, h: int, w: int, noise: float = 1.0 / 255.0) -> np.ndarray:
    """(n_frames, 3, h, w) float32 in [0, 1]: frame t = frame 0 shifted by (2t, -t) px."""
    g = np.random.default_rng(seed)
    base = _smooth_field(g, 3, h, w)
    out = np.empty((n_frames, 3, h, w), np.float32)
    yy = np.arange(h)
    xx = np.arange(w)
    for t in range(n_frames):
        sy = np.clip(yy + t, 0, h - 1)
        sx = np.clip(xx - 2 * t, 0, w - 1)
        f = base[:, sy][:, :, sx]
        f = f + noise * g.standard_normal(f.shape).astype(np.float32)
        out[t] = np.clip(f, 0.0, 1.0)
    return out
