"""Minimal in-memory whole-slide reader (the part of reference ``wsicore/wsireader.py`` the engines
need: ``VirtualWSIReader`` :3121-3694, ``slide_thumbnail``, ``tissue_mask`` :1735-1786).

File-format readers (OpenSlide, TIFF, DICOM, ...) are out of scope: a slide is an
``H x W x 3`` uint8 array (NumPy or CUDA tensor) with an objective power / mpp.  Reads happen on the
GPU: the level-0 image lives in HBM (a 20k x 20k slide is 1.2 GB of the 288 GB) and patches are
gathered from a 255-padded copy, so out-of-bounds regions are white exactly as
``WSIPatchDataset.__getitem__`` pads them (``dataset_abc.py:430-436``).
"""

from __future__ import annotations

import numpy as np
import torch

from tiatoolbox_amd.utils import _tensors


class ArrayWSIReader:
    """ndarray-backed slide at a single (baseline) resolution."""

    def __init__(self, img, mpp: float | None = 0.25, power: float | None = 40.0, mode: str = "rgb") -> None:
        if isinstance(img, torch.Tensor):
            self._dev = img if img.is_cuda else img.to(_tensors.default_device())
        else:
            self._dev = torch.from_numpy(np.ascontiguousarray(img)).to(_tensors.default_device())
        self.mode = mode
        self.mpp = mpp
        self.power = power
        self._padded = None
        self._pad = (0, 0, 0, 0)

    @property
    def img(self) -> np.ndarray:
        return self._dev.cpu().numpy()

    @property
    def device_image(self) -> torch.Tensor:
        return self._dev

    @property
    def slide_dimensions(self) -> tuple[int, int]:
        return int(self._dev.shape[1]), int(self._dev.shape[0])  # (width, height)

    # ------------------------------------------------------------------------------- reads
    def prepare_padding(self, left: int, top: int, right: int, bottom: int, value: int = 255) -> None:
        """Build the padded device copy that makes every later bounded read a plain slice."""
        if self._padded is not None and self._pad == (left, top, right, bottom):
            return
        h, w = self._dev.shape[:2]
        rest = self._dev.shape[2:]
        padded = torch.full((h + top + bottom, w + left + right, *rest), value, dtype=self._dev.dtype,
                            device=self._dev.device)
        padded[top:top + h, left:left + w] = self._dev
        self._padded, self._pad = padded, (left, top, right, bottom)

    def read_bounds_batch(self, bounds: np.ndarray) -> torch.Tensor:
        """Stack of equally-sized regions ``[x0, y0, x1, y1]`` (baseline pixels), padded with 255."""
        bounds = np.asarray(bounds)
        left = max(0, int(-bounds[:, 0].min()))
        top = max(0, int(-bounds[:, 1].min()))
        w, h = self.slide_dimensions
        right = max(0, int(bounds[:, 2].max()) - w)
        bottom = max(0, int(bounds[:, 3].max()) - h)
        pl, pt, pr, pb = self._pad
        if self._padded is None or left > pl or top > pt or right > pr or bottom > pb:
            self.prepare_padding(max(left, pl), max(top, pt), max(right, pr), max(bottom, pb))
        pl, pt, _, _ = self._pad
        return torch.stack([self._padded[y0 + pt:y1 + pt, x0 + pl:x1 + pl] for x0, y0, x1, y1 in bounds.tolist()])

    def read_bounds(self, bounds) -> np.ndarray:
        return self.read_bounds_batch(np.asarray(bounds)[None])[0].cpu().numpy()

    # ---------------------------------------------------------------------- thumbnail / mask
    def slide_thumbnail(self, resolution: float = 1.25, units: str = "power") -> torch.Tensor:
        """Area-averaged down-sample to the requested objective power (integer factor), uint8."""
        if units != "power" or self.power is None:
            msg = "ArrayWSIReader thumbnails are requested by objective power."
            raise ValueError(msg)
        factor = max(1, int(round(self.power / resolution)))
        h, w = self._dev.shape[:2]
        th, tw = h // factor, w // factor
        x = self._dev[:th * factor, :tw * factor].reshape(th, factor, tw, factor, -1).to(torch.float32)
        return torch.round(x.mean(dim=(1, 3))).to(torch.uint8)

    def tissue_mask(self, method: str = "otsu", resolution: float = 1.25, units: str = "power", **masker_kwargs):
        """Tissue mask reader from the thumbnail (ref. ``wsireader.py:1735-1786``)."""
        from tiatoolbox_amd.tools import tissuemask

        thumb = self.slide_thumbnail(resolution, units)
        if method not in ("otsu", "morphological"):
            msg = f"Invalid tissue masking method: {method}."
            raise ValueError(msg)
        if method == "otsu":
            masker = tissuemask.OtsuTissueMasker(**masker_kwargs)
        else:
            masker = tissuemask.MorphologicalMasker(**({"power": resolution} | masker_kwargs))
        mask = masker.fit_transform(thumb[None])[0]
        return ArrayWSIReader(mask.to(torch.uint8) if isinstance(mask, torch.Tensor) else mask.astype(np.uint8),
                              mpp=None, power=resolution, mode="bool")
