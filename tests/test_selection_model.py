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
