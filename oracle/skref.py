"""Restatements of the scikit-image primitives the reference's hot path calls.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  scikit-image (``>=0.26.0``,
reference ``requirements/requirements.txt:27``) is not under ``/root/reference`` and is
not installable here; these follow its published algorithms.
"""

from __future__ import annotations

import numpy as np


def rescale_intensity(image: np.ndarray, in_range: tuple, out_range: tuple) -> np.ndarray:
    """``skimage.exposure.rescale_intensity(image, in_range=(lo,hi), out_range=(a,b))``.

    With an explicit tuple ``out_range`` the output dtype is float64; arithmetic is
    ``clip -> (x-imin)/(imax-imin) -> x*(omax-omin)+omin`` in float64.  Pinned by the
    reference golden ``tests/test_utils.py:882-911`` (via ``contrast_enhancer``).
    Call site: ``tiatoolbox/utils/misc.py:439-443``.
    """
    imin, imax = float(in_range[0]), float(in_range[1])
    omin, omax = float(out_range[0]), float(out_range[1])
    x = np.clip(np.asarray(image, dtype=np.float64), imin, imax)
    if imin != imax:
        x = (x - imin) / (imax - imin)
        return x * (omax - omin) + omin
    return np.clip(x, omin, omax)


def threshold_otsu_u8(image: np.ndarray) -> int:
    """``skimage.filters.threshold_otsu`` for an integer (uint8) image.

    ``histogram(image, nbins=256, source_range='image')`` on integer input yields one
    bin per integer in ``[min, max]`` (bin centres = the integers); threshold = centre
    of ``argmax(w1[:-1] * w2[1:] * (mean1[:-1] - mean2[1:])**2)``.  If the image has a
    single value that value is returned.  Call site: ``tiatoolbox/tools/tissuemask.py:134``.
    """
    image = np.asarray(image)
    first = image.reshape(-1)[0]
    if np.all(image == first):
        return int(first)
    lo, hi = int(image.min()), int(image.max())
    counts = np.bincount(image.ravel().astype(np.int64) - lo, minlength=hi - lo + 1).astype(np.float64)
    centers = np.arange(lo, hi + 1, dtype=np.float64)
    weight1 = np.cumsum(counts)
    weight2 = np.cumsum(counts[::-1])[::-1]
    mean1 = np.cumsum(counts * centers) / weight1
    mean2 = (np.cumsum((counts * centers)[::-1]) / weight2[::-1])[::-1]
    variance12 = weight1[:-1] * weight2[1:] * (mean1[:-1] - mean2[1:]) ** 2
    idx = int(np.argmax(variance12))
    return int(centers[idx])


def remove_small_objects_labels(lab: np.ndarray, max_size: int) -> np.ndarray:
    """``skimage.morphology.remove_small_objects(label_image, max_size=s)``.

    Zeroes every label whose pixel count is ``<= max_size`` (no relabelling).  Call
    sites: ``tiatoolbox/models/architecture/hovernet.py:544,614``.
    """
    out = lab.copy()
    counts = np.bincount(out.ravel())
    small = counts <= max_size
    small[0] = False
    out[small[out]] = 0
    return out


def remove_small_objects(ar: np.ndarray, max_size: int) -> np.ndarray:
    """``skimage.morphology.remove_small_objects(ar, max_size=s)`` for both input kinds: a boolean image is first
    labelled with connectivity 1 (``ndimage.label`` with its default cross) and returned boolean; an integer image
    is taken as labels.  Objects with ``area <= max_size`` go.  Call sites: ``hovernet.py:544,614`` (labels),
    ``hovernetplus.py:162-165`` (boolean)."""
    ar = np.asarray(ar)
    if ar.dtype == bool:
        from scipy import ndimage

        lab = ndimage.label(ar)[0]
        return remove_small_objects_labels(lab, max_size) > 0
    return remove_small_objects_labels(ar, max_size)


class _Heap:
    """skimage's ``heap_general.pxi`` binary min-heap, ordered by ``(value, age)`` only.

    ``push``: append, then swap with the parent while *smaller* than it.  ``pop``: the last item replaces the root, then
    sifts down -- at every node the smallest of (node, left child, right child) moves up, stopping as soon as the node
    itself is the smallest.  (Not CPython's ``heapq`` procedure, which first bubbles the smaller child up to a leaf; the
    two only differ in the order in which entries that TIE on ``(value, age)`` leave the queue -- e.g. the initial
    markers of a plateau, which all carry age 0.)  Restated from memory of scikit-image 0.2x; parity unpinned.
    """

    __slots__ = ("items",)

    def __init__(self) -> None:
        self.items: list[tuple[float, int, int]] = []

    @staticmethod
    def _smaller(a: tuple, b: tuple) -> bool:
        if a[0] != b[0]:
            return a[0] < b[0]
        return a[1] < b[1]

    def push(self, item: tuple[float, int, int]) -> None:
        a = self.items
        a.append(item)
        child = len(a) - 1
        while child > 0:
            parent = (child + 1) // 2 - 1
            if not self._smaller(a[child], a[parent]):
                break
            a[child], a[parent] = a[parent], a[child]
            child = parent

    def pop(self) -> tuple[float, int, int]:
        a = self.items
        top = a[0]
        last = a.pop()
        if not a:
            return top
        a[0] = last
        i, n = 0, len(a)
        while True:
            left, right, smallest = 2 * i + 1, 2 * i + 2, i
            if left >= n:
                break
            if self._smaller(a[left], a[i]):
                smallest = left
            if right < n and self._smaller(a[right], a[smallest]):
                smallest = right
            if smallest == i:
                break
            a[i], a[smallest] = a[smallest], a[i]
            i = smallest
        return top


def watershed(image: np.ndarray, markers: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """``skimage.segmentation.watershed(image, markers, mask=mask)`` (connectivity 1).

    Priority flood (``_watershed_cy.watershed_raveled``): a binary heap (:class:`_Heap`) ordered by ``(value, age)``;
    every marker pixel (raster order) is pushed with its own value and age 0; a popped pixel visits its neighbours in
    raveled offset order (up, left, right, down); each unlabelled in-mask neighbour takes the popped pixel's label and is
    pushed with its own image value and the next age.  Call site:
    ``tiatoolbox/models/architecture/hovernet.py:616``.
    """
    image = np.asarray(image, dtype=np.float64)
    h, w = image.shape
    out = np.where(mask, markers, 0).astype(np.int32)
    maskb = np.asarray(mask, dtype=bool)
    heap = _Heap()
    age = 0
    flat_out = out.ravel()
    flat_img = image.ravel()
    flat_mask = maskb.ravel()
    for idx in np.flatnonzero(flat_out):
        heap.push((float(flat_img[idx]), 0, int(idx)))
    while heap.items:
        idx = heap.pop()[2]
        r, c = divmod(idx, w)
        lab = flat_out[idx]
        for dr, dc in ((-1, 0), (0, -1), (0, 1), (1, 0)):
            rr, cc = r + dr, c + dc
            if rr < 0 or rr >= h or cc < 0 or cc >= w:
                continue
            n = rr * w + cc
            if flat_out[n] != 0 or not flat_mask[n]:
                continue
            age += 1
            flat_out[n] = lab
            heap.push((float(flat_img[n]), age, n))
    return out
