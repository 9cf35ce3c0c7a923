"""Model check of the watershed-by-relaxation argument behind ``ws_relax_wave_kernel`` (csrc/hover_post.hip).

The device kernel labels every blob by a parallel minimax relaxation and hands a blob to the sequential heap flood only
when its labelling hinges on an exact tie.  This file restates the relaxation in NumPy (whole image at once, Jacobi
sweeps) and checks, against the oracle's priority flood (``oracle/skref.watershed`` = scikit-image's algorithm), that

* every blob the relaxation does NOT flag gets exactly the oracle's labels -- on generic fields, on heavily quantised
  fields (many exact ties) and on plateaus;
* on HoVer-Net-like maps no blob is flagged at all (so the fast path is the one that runs).

CPU only; the HIP kernel itself is compared with the oracle in ``tests/test_hovernet_post.py``.
"""

from __future__ import annotations

import numpy as np
from scipy import ndimage

from oracle import hovernet as ohv
from oracle import skref

OFFS = ((-1, 0), (0, -1), (0, 1), (1, 0))


def _shift(a: np.ndarray, dy: int, dx: int, fill) -> np.ndarray:
    h, w = a.shape
    r = np.full_like(a, fill)
    r[max(-dy, 0):h + min(-dy, 0), max(-dx, 0):w + min(-dx, 0)] = a[max(dy, 0):h + min(dy, 0), max(dx, 0):w + min(dx, 0)]
    return r


def relax_watershed(image: np.ndarray, markers: np.ndarray, mask: np.ndarray):
    """(labels, conflict mask, sweeps): L = highest value on the claim chain, D = hops since that pixel."""
    image = np.asarray(image, np.float64)
    mask = np.asarray(mask, bool)
    lab = np.where(mask, markers, 0).astype(np.int64)
    unl = mask & (lab == 0)
    big = np.inf
    lev = np.where(lab > 0, image, big)
    hops = np.zeros(image.shape, np.int64)
    sweeps = 0
    while True:
        sweeps += 1
        b_l = np.full(image.shape, big)
        b_d = np.zeros(image.shape, np.int64)
        b_lab = np.zeros(image.shape, np.int64)
        for dy, dx in OFFS:
            q_l, q_d, q_lab = _shift(lev, dy, dx, big), _shift(hops, dy, dx, 0), _shift(lab, dy, dx, 0)
            better = (q_l < b_l) | ((q_l == b_l) & np.isfinite(q_l) & (q_d < b_d))
            b_l, b_d, b_lab = np.where(better, q_l, b_l), np.where(better, q_d, b_d), np.where(better, q_lab, b_lab)
        up = image > b_l
        n_l, n_d = np.where(up, image, b_l), np.where(up, 0, b_d + 1)
        ch = unl & np.isfinite(b_l) & ((n_l != lev) | (n_d != hops) | (b_lab != lab))
        if not ch.any():
            break
        lev, hops, lab = np.where(ch, n_l, lev), np.where(ch, n_d, hops), np.where(ch, b_lab, lab)
    m_l = np.full(image.shape, big)
    for dy, dx in OFFS:
        m_l = np.minimum(m_l, _shift(lev, dy, dx, big))
    conflict = np.zeros(image.shape, bool)
    for dy, dx in OFFS:
        conflict |= unl & np.isfinite(m_l) & (_shift(lev, dy, dx, big) == m_l) & (_shift(lab, dy, dx, 0) != lab)
    return lab.astype(np.int32), conflict, sweeps


def _compare(image, markers, mask):
    ref = skref.watershed(image, markers, mask)
    got, conflict, _ = relax_watershed(image, markers, mask)
    blobs, n_blobs = ndimage.label(mask)
    flagged = set(np.unique(blobs[conflict]).tolist()) - {0}
    settled = ~np.isin(blobs, list(flagged))
    return int(((ref != got) & settled).sum()), len(flagged), n_blobs


def test_unflagged_blobs_equal_the_priority_flood():
    rng = np.random.default_rng(7)
    blobs = flagged = 0
    for trial in range(48):
        h, w = (int(v) for v in rng.integers(12, 40, 2))
        kind = trial % 4
        if kind == 0:
            img = rng.standard_normal((h, w))
        elif kind == 1:
            img = ndimage.gaussian_filter(rng.standard_normal((h, w)), 2.0)
        elif kind == 2:  # heavily quantised: exact ties everywhere
            img = np.round(ndimage.gaussian_filter(rng.standard_normal((h, w)), 1.5) * 16) / 16
        else:  # plateaus at 0 and -1, like a clipped energy map
            img = -np.clip(ndimage.gaussian_filter(rng.standard_normal((h, w)), 2.0) * 3 + 0.5, 0, 1)
        mask = ndimage.gaussian_filter(rng.standard_normal((h, w)), 2.5) > -0.2
        seeds = (rng.random((h, w)) < 0.025) & mask
        seeds = ndimage.binary_dilation(seeds, iterations=int(rng.integers(0, 3))) & mask
        markers, _ = ndimage.label(seeds)
        bad, n_flagged, n_blobs = _compare(img, markers, mask)
        assert bad == 0, (trial, kind, bad)
        blobs += n_blobs
        flagged += n_flagged
    assert blobs > 40
    assert flagged > 0  # the tie-heavy fields do exercise the flag


def test_tie_free_fields_are_never_flagged():
    rng = np.random.default_rng(3)
    for _ in range(6):
        img = ndimage.gaussian_filter(rng.standard_normal((32, 32)), 1.5)  # distinct float64 values
        assert np.unique(img).size == img.size
        mask = np.ones((32, 32), bool)
        markers, _ = ndimage.label(rng.random((32, 32)) < 0.03)
        bad, n_flagged, _ = _compare(img, markers, mask)
        assert bad == 0 and n_flagged == 0


def test_hovernet_like_maps_need_no_fallback():
    maps = ohv.synth_maps(1, 96, 96, seed=2, n_blobs=18)
    dbg: dict = {}
    ohv.proc_np_hv(maps[0][0], maps[1][0], debug=dbg)
    bad, n_flagged, n_blobs = _compare(dbg["dist"], dbg["marker"], dbg["blb"] > 0)
    assert n_blobs >= 5 and bad == 0 and n_flagged == 0
