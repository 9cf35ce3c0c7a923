"""Tissue maskers (API of reference ``tiatoolbox/tools/tissuemask.py``).

``OtsuTissueMasker``: grey conversion fused with one global 256-bin histogram (one pass over
the pixels), Otsu's threshold from the 256 counts on the device too (``fit`` and ``transform``
enqueue back to back; the ``threshold`` attribute syncs on first read), thresholding fused with
the grey conversion.
``MorphologicalMasker``: + 8-connected component labelling, small-region removal and
elliptical dilation, all on the GPU.  uint8 RGB / single-channel inputs take the HIP path;
other dtypes (the reference's own known-answer test feeds 0/1 floats) are histogrammed with
torch on the device (256 linear bins, as scikit-image does for float images).
"""

from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np
import torch

from tiatoolbox_amd.tools import _img_device as img
from tiatoolbox_amd.utils import _tensors


def objective_power2mpp(objective_power):
    """Reference ``utils/misc.py:346-374``."""
    return 10.0 / np.array(objective_power)


def _otsu_from_counts(counts: np.ndarray, centers: np.ndarray) -> float:
    """Otsu's threshold from a histogram (scikit-image ``threshold_otsu`` arithmetic)."""
    counts = counts.astype(np.float64)
    weight1 = np.cumsum(counts)
    weight2 = np.cumsum(counts[::-1])[::-1]
    with np.errstate(invalid="ignore", divide="ignore"):
        mean1 = np.cumsum(counts * centers) / weight1
        mean2 = (np.cumsum((counts * centers)[::-1]) / weight2[::-1])[::-1]
    variance12 = weight1[:-1] * weight2[1:] * (mean1[:-1] - mean2[1:]) ** 2
    return centers[int(np.nanargmax(variance12))]


class TissueMasker(ABC):
    """Tissue masker base class (ref. :14-72)."""

    @abstractmethod
    def fit(self, images, masks=None) -> None:
        ...

    @abstractmethod
    def transform(self, images):
        ...

    def fit_transform(self, images, **kwargs):
        self.fit(images, masks=None, **kwargs)
        return self.transform(images)


def _to_device_images(images) -> tuple[torch.Tensor, bool]:
    """List / array of images -> one device tensor [N,H,W,C]; flag = result as NumPy."""
    if isinstance(images, torch.Tensor):
        t = images if images.is_cuda else images.to(_tensors.default_device())
        return t, False
    arr = np.asarray(images)
    return torch.from_numpy(np.ascontiguousarray(arr)).to(_tensors.default_device()), True


class OtsuTissueMasker(TissueMasker):
    """Otsu threshold over all pixels of all images (ref. :75-164)."""

    def __init__(self) -> None:
        super().__init__()
        self._threshold = None
        self._threshold_dev: torch.Tensor | None = None
        self.fitted = False

    @property
    def threshold(self):
        """The fitted threshold (ref. attribute ``self.threshold``).  A uint8 ``fit`` leaves it on the device (``fit`` and
        ``transform`` enqueue back to back); reading the attribute brings the integer to the host once."""
        if self._threshold is None and self._threshold_dev is not None:
            self._threshold = int(self._threshold_dev[0].item())
        return self._threshold

    @threshold.setter
    def threshold(self, value) -> None:
        self._threshold = value
        self._threshold_dev = None

    def fit(self, images, masks=None) -> None:  # noqa: ARG002
        images_shape = tuple(images.shape) if isinstance(images, torch.Tensor) else np.shape(images)
        if len(images_shape) != 4:  # noqa: PLR2004
            msg = (f"Expected 4 dimensional input shape (N, height, width, 3) "
                   f"but received shape of {images_shape}.")
            raise ValueError(msg)
        t, _ = _to_device_images(images)
        if t.dtype == torch.uint8:
            # one pass over the pixels (grey + histogram fused), Otsu's arithmetic on the 256 counts on the device
            rgb = t.shape[-1] == 3  # noqa: PLR2004
            self._threshold = None
            self._threshold_dev = img.otsu_fit(t if rgb else t[..., 0].contiguous(), channels=3 if rgb else 1)
        else:
            grey = t[..., 0].to(torch.float64)
            lo, hi = float(grey.min()), float(grey.max())
            if lo == hi:
                self.threshold = lo
            else:
                counts = torch.histc(grey, bins=256, min=lo, max=hi).cpu().numpy()
                edges = np.linspace(lo, hi, 257)
                self.threshold = float(_otsu_from_counts(counts, (edges[:-1] + edges[1:]) / 2.0))
        self.fitted = True

    def _masks(self, t: torch.Tensor, dtype: torch.dtype = torch.uint8) -> torch.Tensor:
        if t.dtype == torch.uint8:
            is_rgb = t.dim() == 4 and t.shape[-1] == 3  # noqa: PLR2004
            src = t if is_rgb else (t[..., 0].contiguous() if t.dim() == 4 else t)  # noqa: PLR2004
            if self._threshold is None and self._threshold_dev is not None:
                return img.threshold_lt(src, self._threshold_dev, is_rgb=is_rgb, dtype=dtype)
            # grey < threshold with an integer grey: equivalent integer bound
            thr = int(np.ceil(self.threshold)) if float(self.threshold) != int(self.threshold) else int(self.threshold)
            return img.threshold_lt(src, thr, is_rgb=is_rgb, dtype=dtype)
        grey = t[..., 0] if t.dim() == 4 else t  # noqa: PLR2004
        return (grey < self.threshold).to(dtype)

    def transform(self, images):
        if not self.fitted:
            msg = "Fit must be called before transform."
            raise SyntaxError(msg)
        t, as_numpy = _to_device_images(images)
        masks = self._masks(t, torch.bool)  # the kernel's 0 / 1 bytes are a bool tensor's storage
        return masks.cpu().numpy() if as_numpy else masks


class MorphologicalMasker(OtsuTissueMasker):
    """Otsu + small-region removal (8-connected) + elliptical dilation (ref. :167-306)."""

    def __init__(self, *, mpp=None, power=None, kernel_size=None, min_region_size=None) -> None:
        """Exactly one of ``mpp`` / ``power`` / ``kernel_size`` (or none: a 1x1 element) sizes the ellipse.

        Contract of the reference constructor (``tissuemask.py:229-272``): ``power`` is mapped to microns per pixel,
        an mpp of ``m`` gives an element of ``max(32 / m, 1)`` pixels per axis (the code's 32, not the docstring's
        64), scalars are broadcast to both axes, sizes are rounded half-to-even, and ``min_region_size`` defaults to
        the number of set pixels of the element.
        """
        super().__init__()
        given = {name: val for name, val in (("mpp", mpp), ("power", power), ("kernel_size", kernel_size))
                 if val is not None}
        if len(given) > 1:
            msg = "Only one of mpp, power, kernel_size can be given."
            raise ValueError(msg)

        def pair(value) -> np.ndarray:
            arr = np.asarray(value, dtype=np.float64).reshape(-1)
            return np.repeat(arr, 2) if arr.size == 1 else arr

        if "power" in given:
            given = {"mpp": objective_power2mpp(power)}
        if "mpp" in given:
            size = np.maximum(32 / pair(given["mpp"]), 1.0)
        else:
            size = pair(given.get("kernel_size", 1))
        self.kernel_size = tuple(np.round(size).astype(int))
        self.kernel = img.get_structuring_element_ellipse(self.kernel_size)
        self.min_region_size = int(self.kernel.sum()) if min_region_size is None else min_region_size

    def transform(self, images):
        if not self.fitted:
            msg = "Fit must be called before transform."
            raise SyntaxError(msg)
        t, as_numpy = _to_device_images(images)
        out = None
        if t.dtype == torch.uint8 and t.dim() == 4 and t.shape[-1] in (1, 3):  # noqa: PLR2004
            # one launch: threshold, small-region removal and dilation tile by tile in LDS (element + min_region_size within the halo)
            rgb = t.shape[-1] == 3  # noqa: PLR2004
            if self._threshold is None and self._threshold_dev is not None:
                thr = self._threshold_dev
            else:
                thr = int(np.ceil(self.threshold)) if float(self.threshold) != int(self.threshold) else int(self.threshold)
            out = img.morph_mask(t if rgb else t[..., 0].contiguous(), thr, int(self.min_region_size), self._offsets(t.device),
                                 channels=3 if rgb else 1, reach=self._reach())
        if out is None:
            mask = self._masks(t)
            labels, _ = img.ccl_label(mask, connectivity=8)
            img.label_area_filter(labels, int(self.min_region_size))
            keep = (labels > 0).to(torch.uint8)
            out = img.binary_morph(keep, self._offsets(keep.device), "dilate").bool()
        return out.cpu().numpy() if as_numpy else out

    def _reach(self) -> int:
        """Largest |dy|, |dx| of the element around its anchor (host arithmetic: no device round trip per call)."""
        kh, kw = self.kernel.shape
        ys, xs = np.nonzero(self.kernel)
        return int(max(np.abs(ys - kh // 2).max(initial=0), np.abs(xs - kw // 2).max(initial=0)))

    def _offsets(self, device: torch.device) -> torch.Tensor:
        """(dy, dx) offsets of the element on ``device`` (cached: a few dozen int32)."""
        key = str(device)
        cache = self.__dict__.setdefault("_offsets_dev", {})
        if key not in cache:
            cache[key] = img.offsets_of(self.kernel, device)
        return cache[key]
