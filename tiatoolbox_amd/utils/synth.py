"""Seeded synthetic inputs for tests and ``bench.py``.

No network and no real slides here, so the benchmark workloads are synthetic:

* ``g_uniform``: uniform random bytes (stress case).
* ``g_he``: H&E-like patches.  Per patch a stain matrix (Ruifrok rows + N(0,0.02),
  re-normalised), Gamma-distributed haematoxylin / eosin concentrations smoothed by a
  5x5 box, ``OD = C.S + N(0,0.01)``, 20 % of pixels replaced by bright background,
  ``RGB = uint8(clip(255*exp(-OD)))``.  Guarantees a non-empty tissue mask and a
  non-degenerate eigen-gap for Macenko.
"""

from __future__ import annotations

import numpy as np

_RUIFROK = np.array([[0.65, 0.70, 0.29], [0.07, 0.99, 0.11]])


def g_uniform(n: int, h: int, w: int, seed: int = 0) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)


def _box5(x: np.ndarray) -> np.ndarray:
    """5x5 box filter over the last two axes of (n,h,w), edge-replicated."""
    p = np.pad(x, ((0, 0), (2, 2), (2, 2)), mode="edge")
    c = np.cumsum(p, axis=1)
    c = np.concatenate([np.zeros_like(c[:, :1]), c], axis=1)
    r = c[:, 5:] - c[:, :-5]
    c = np.cumsum(r, axis=2)
    c = np.concatenate([np.zeros_like(c[:, :, :1]), c], axis=2)
    return (c[:, :, 5:] - c[:, :, :-5]) / 25.0


def g_he(n: int, h: int, w: int, seed: int = 1, chunk: int = 64) -> np.ndarray:
    rng = np.random.default_rng(seed)
    out = np.empty((n, h, w, 3), dtype=np.uint8)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        sm = _RUIFROK[None] + rng.normal(0.0, 0.02, (m, 2, 3))
        sm = np.abs(sm)
        sm /= np.linalg.norm(sm, axis=2, keepdims=True)
        ch = _box5(rng.gamma(2.0, 0.35, (m, h, w)))
        ce = _box5(rng.gamma(2.0, 0.25, (m, h, w)))
        od = ch[..., None] * sm[:, None, None, 0, :] + ce[..., None] * sm[:, None, None, 1, :]
        od += rng.normal(0.0, 0.01, od.shape)
        rgb = 255.0 * np.exp(-np.maximum(od, 0.0))
        bg = rng.random((m, h, w)) < 0.2
        bgv = rng.uniform(225, 255, (m, h, w, 3))
        rgb = np.where(bg[..., None], bgv, rgb)
        out[s:s + m] = np.clip(rgb, 0, 255).astype(np.uint8)
    return out


def hover_head_maps(n: int, h: int, w: int, seed: int = 0, n_blobs: int = 30, num_types: int = 6):
    """Synthetic HoVer-Net head outputs: ``np`` = union of Gaussian blobs,
    ``hv`` = per-blob normalised x/y ramps in [-1, 1] plus noise, ``tp`` = blob class."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    np_map = np.zeros((n, h, w, 1), np.float32)
    hv = np.zeros((n, h, w, 2), np.float32)
    tp = np.zeros((n, h, w, 1), np.float32)
    for i in range(n):
        best = np.zeros((h, w), np.float32)
        for _ in range(n_blobs):
            cy, cx = rng.uniform(4, h - 4), rng.uniform(4, w - 4)
            ry, rx = rng.uniform(3.5, 9.0, 2)
            d = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2
            p = np.exp(-0.5 * d * 2.0).astype(np.float32)
            upd = p > best
            best = np.where(upd, p, best)
            hv[i, ..., 0] = np.where(upd, np.clip((xx - cx) / rx, -1, 1), hv[i, ..., 0])
            hv[i, ..., 1] = np.where(upd, np.clip((yy - cy) / ry, -1, 1), hv[i, ..., 1])
            tp[i, ..., 0] = np.where(upd & (p > 0.3), rng.integers(1, num_types), tp[i, ..., 0])
        np_map[i, ..., 0] = best
        hv[i] *= (best > 0.2)[..., None]
    np_map += rng.normal(0, 0.02, np_map.shape).astype(np.float32)
    hv += rng.normal(0, 0.02, hv.shape).astype(np.float32)
    return np.clip(np_map, 0, 1).astype(np.float32), hv.astype(np.float32), tp
