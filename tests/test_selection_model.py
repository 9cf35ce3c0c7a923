"""Model check of the selection scheme behind the experimental ``-DTIA_F32_BINS=1`` build of ``stain_stats.hip``
(off by default): exact order statistics from a histogram of APPROXIMATE keys.

Claim: if every approximate key is within ``eps < bin_width / 4`` of its exact key and ``b`` is the bin that holds rank
``r`` of the approximate keys, then (1) every element in bins ``<= b-2`` is exactly smaller and every element in bins
``>= b+2`` exactly larger than the exact r-th value, and (2) the exact (r+1)-th value lies in bins ``b-1..b+1`` or in
``{b2, b2+1}`` with ``b2`` the first non-empty bin ``>= b+2``.  Hence sorting the exact keys of those bins and indexing
with ``r - count(bins <= b-2)`` yields the exact r-th and (r+1)-th values.  The kernel falls back to the all-f64 path
whenever a precondition fails (edge bins, too many candidates)."""

from __future__ import annotations

import numpy as np


def _select(exact, approx, r, lo, delta, nb, cap=1024):
    n = len(exact)
    bins = np.clip(np.floor((approx - lo) / delta), 0, nb - 1).astype(int)
    counts = np.bincount(bins, minlength=nb)
    cum = np.cumsum(counts)
    b = int(np.searchsorted(cum, r, side="right"))
    if b - 1 < 1 or b + 1 > nb - 2:
        return None
    cand = (bins >= b - 1) & (bins <= b + 1)
    beyond = np.flatnonzero(counts[b + 2:])
    if len(beyond):
        b2 = b + 2 + int(beyond[0])
        if b2 + 1 > nb - 2:
            return None
        cand |= (bins >= b2) & (bins <= b2 + 1)
    if cand.sum() > cap:
        return None
    srt = np.sort(exact[cand])
    rr = r - cum[b - 2]
    assert 0 <= rr < len(srt)
    if r + 1 >= n:
        return srt[rr], srt[rr]
    assert rr + 1 < len(srt)
    return srt[rr], srt[rr + 1]


def test_exact_order_statistics_from_approximate_bins():
    rng = np.random.default_rng(0)
    accepted = fallbacks = 0
    for trial in range(1500):
        n = int(rng.integers(50, 5000))
        kind = trial % 4
        if kind == 0:
            exact = rng.normal(0, 1, n)
        elif kind == 1:
            exact = rng.gamma(2, 0.3, n)
        elif kind == 2:
            exact = np.round(rng.normal(0, 1, n), 2)  # many exact ties
        else:
            exact = np.concatenate([rng.normal(0, 1, n // 2), rng.normal(5, 0.01, n - n // 2)])
        lo, hi = exact.mean() - 8 * exact.std(), exact.mean() + 12 * exact.std()
        nb = 4096
        delta = (hi - lo) / nb
        eps = delta * rng.uniform(0.0, 0.249)
        noise = rng.uniform(-eps, eps, n)
        if trial % 3 == 0:  # adversarial: push values that sit near a bin edge across it
            edge = lo + np.round((exact - lo) / delta) * delta
            near = np.abs(exact - edge) < eps
            noise[near] = np.where(edge[near] >= exact[near], 1.0, -1.0) * eps * 0.999
        srt = np.sort(exact)
        for q in (0.01, 0.5, 0.99):
            r = int(np.floor((n - 1) * q))
            res = _select(exact, exact + noise, r, lo, delta, nb)
            if res is None:
                fallbacks += 1
                continue
            accepted += 1
            assert res[0] == srt[r] and res[1] == srt[min(r + 1, n - 1)]
    assert accepted > 10 * fallbacks


def test_float32_pseudo_angle_error_model():
    """The per-pixel error bound the experimental kernel relies on: float32 projections (fma chain on the float32 OD
    table), reciprocal pessimised by one ulp, quadrant arithmetic in float32 -- against the float64 key.  Bound:
    ``4.8e-7 * (od_r+od_g+od_b) / (|p0|+|p1|) + 5e-7``; pixels with ``200 (|p0|+|p1|) <= l1`` would take the float64 key.
    On stained-tissue data the bound holds everywhere, stays below 1 % of a bin, and no pixel is 'uncertain'."""
    from pathlib import Path

    from oracle import stain as ostain
    from tiatoolbox_amd.utils import cvtables, synth

    f32 = np.float32

    def fma32(a, b, c):
        return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(f32)

    od64 = cvtables.od_lut()
    od32 = od64.astype(f32)
    delta = 4.001953125 / 4096
    target = np.load(Path(__file__).parent / "golden" / "target_crop_256.npy")
    for patch in [*synth.g_he(3, 224, 224, seed=1), target]:
        mask = ostain.get_luminosity_tissue_mask(patch.copy(), threshold=0.8).reshape(-1)
        px = patch.reshape(-1, 3)[mask]
        o64, o32 = od64[px], od32[px]
        _, vec = np.linalg.eigh(np.cov(o64, rowvar=False))
        e1, e2 = vec[:, 2] * np.sign(vec[0, 2] or 1), vec[:, 1] * np.sign(vec[0, 1] or 1)
        p0, p1 = o64 @ e1, o64 @ e2
        d = np.abs(p0) + np.abs(p1)
        r = p1 / d
        k64 = np.where(p0 >= 0, r, np.where(p1 >= 0, 2 - r, -2 - r))
        q = []
        for e in (e1.astype(f32), e2.astype(f32)):
            q.append(fma32(o32[:, 2], e[2], fma32(o32[:, 1], e[1], (o32[:, 0] * e[0]).astype(f32))))
        ds = (np.abs(q[0]) + np.abs(q[1])).astype(f32)
        rcp = np.nextafter((f32(1) / ds).astype(f32), f32(np.inf))
        rq = (q[1] * rcp).astype(f32)
        k32 = np.where(q[0] >= 0, rq, np.where(q[1] >= 0, (f32(2) - rq).astype(f32), (f32(-2) - rq).astype(f32)))
        l1 = o32.sum(1)
        certain = 200.0 * ds > l1
        err = np.abs(k32.astype(np.float64) - k64)
        bound = 4.8e-7 * l1 / d + 5e-7
        assert certain.all()
        assert (err <= bound).all() and bound.max() < 0.01 * delta
