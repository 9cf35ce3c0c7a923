"""Minimal in-memory whole-slide reader (the part of reference ``wsicore/wsireader.py`` the engines
need: ``VirtualWSIReader`` :3121-3694, ``slide_thumbnail``, ``tissue_mask`` :1735-1786).

File-format readers (OpenSlide, TIFF, DICOM, ...) are out of scope: a slide is an
``H x W x 3`` uint8 array (NumPy or CUDA tensor) with an objective power / mpp.  Reads happen on the
GPU: the level-0 image lives in HBM (a 20k x 20k slide is 1.2 GB of the 288 GB) and a batch of patches is
gathered by one kernel launch (``tia_gather_patches_u8``) that writes 255 wherever a region leaves the slide, exactly
as ``WSIPatchDataset.__getitem__`` pads (``dataset_abc.py:430-436``).
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
    def read_bounds_batch(self, bounds, pad_value: int = 255, *, size: tuple[int, int] | None = None) -> torch.Tensor:
        """Stack of equally-sized regions ``[x0, y0, x1, y1]`` (baseline pixels), ``pad_value`` outside the slide:
        one gather launch (``tia_gather_patches_u8``) writes the whole ``[M, ph, pw, C]`` batch.

        ``bounds`` may already be an ``int32 [M, 4]`` tensor on the slide's device (with ``size=(pw, ph)``): the WSI loops
        upload every patch's bounds once and pass slices, so that no batch waits on a host -> device copy."""
        from tiatoolbox_amd import _lib

        if isinstance(bounds, torch.Tensor) and bounds.is_cuda:
            if size is None or bounds.dtype != torch.int32 or bounds.dim() != 2 or bounds.shape[1] != 4 or not bounds.is_contiguous():  # noqa: PLR2004
                msg = "device bounds: a contiguous int32 [M, 4] tensor together with size=(pw, ph)."
                raise ValueError(msg)
            src = self._dev if self._dev.dim() == 3 else self._dev[..., None]  # noqa: PLR2004
            pw, ph = int(size[0]), int(size[1])
            if src.dtype != torch.uint8 or not src.is_contiguous() or (ph * pw * src.shape[2]) % 4 != 0 or len(bounds) > 65535:  # noqa: PLR2004
                return self.read_bounds_batch(bounds.cpu().numpy(), pad_value)
            sh, sw, c = src.shape
            out = torch.empty((len(bounds), ph, pw, c), dtype=torch.uint8, device=src.device)
            with torch.cuda.device(src.device):
                rc = _lib.load().tia_gather_patches_u8(src.data_ptr(), sh, sw, c, bounds.data_ptr(), len(bounds), ph, pw, int(pad_value),
                                                       out.data_ptr(), _lib.current_stream())
            _lib.check(rc, "tia_gather_patches_u8")
            return out if self._dev.dim() == 3 else out[..., 0]  # noqa: PLR2004
        bounds = np.ascontiguousarray(np.asarray(bounds).reshape(-1, 4), dtype=np.int32)
        sizes = np.unique(np.stack([bounds[:, 2] - bounds[:, 0], bounds[:, 3] - bounds[:, 1]], axis=1), axis=0)
        if len(sizes) != 1:
            msg = "read_bounds_batch expects regions of one size."
            raise ValueError(msg)
        pw, ph = int(sizes[0, 0]), int(sizes[0, 1])
        src = self._dev if self._dev.dim() == 3 else self._dev[..., None]  # noqa: PLR2004
        src = src.contiguous()
        if src.dtype == torch.bool:
            src = src.to(torch.uint8)
        if src.dtype != torch.uint8:
            msg = "device patch reads need a uint8 slide."
            raise TypeError(msg)
        sh, sw, c = src.shape
        m = len(bounds)
        out = torch.empty((m, ph, pw, c), dtype=torch.uint8, device=src.device)
        if (ph * pw * c) % 4 != 0:  # odd-sized patches: plain slicing of a padded copy (rare; keeps the contract)
            pad = max(0, -int(bounds[:, :2].min()), int(bounds[:, 2].max()) - sw, int(bounds[:, 3].max()) - sh)
            padded = torch.nn.functional.pad(src.permute(2, 0, 1), (pad, pad, pad, pad), value=pad_value).permute(1, 2, 0)
            for i, (x0, y0, x1, y1) in enumerate(bounds.tolist()):
                out[i] = padded[y0 + pad:y1 + pad, x0 + pad:x1 + pad]
        else:
            bt = torch.from_numpy(bounds).to(src.device)
            lib = _lib.load()
            with torch.cuda.device(src.device):
                for s in range(0, m, 65535):
                    k = min(65535, m - s)
                    rc = lib.tia_gather_patches_u8(src.data_ptr(), sh, sw, c, bt[s:s + k].data_ptr(), k, ph, pw, int(pad_value),
                                                   out[s:s + k].data_ptr(), _lib.current_stream())
                    _lib.check(rc, "tia_gather_patches_u8")
        return out if self._dev.dim() == 3 else out[..., 0]  # noqa: PLR2004

    def read_bounds(self, bounds) -> np.ndarray:
        return self.read_bounds_batch(np.asarray(bounds)[None])[0].cpu().numpy()

    # ---------------------------------------------------------------------- thumbnail / mask
    def slide_thumbnail(self, resolution: float = 1.25, units: str = "power") -> torch.Tensor:
        """Thumbnail at the requested objective power, uint8.  The reference reads it through ``imresize`` whose
        down-sampling interpolation is ``cv2.INTER_AREA`` (``utils/transforms.py:imresize``, ``wsireader.py:1735-1786``);
        for an integer factor that is the exact box mean rounded half-to-even (``cvRound``), which is what this does."""
        if units != "power" or self.power is None:
            msg = "ArrayWSIReader thumbnails are requested by objective power."
            raise ValueError(msg)
        ratio = self.power / resolution
        factor = max(1, int(round(ratio)))
        if abs(ratio - factor) > 1e-9:  # noqa: PLR2004
            msg = (f"ArrayWSIReader holds one (baseline) level: thumbnails need an integer down-sampling factor, got "
                   f"{self.power}/{resolution}.")
            raise ValueError(msg)
        h, w = self._dev.shape[:2]
        th, tw = h // factor, w // factor
        if th == 0 or tw == 0:
            msg = f"the slide ({h} x {w}) is smaller than one thumbnail pixel at factor {factor}."
            raise ValueError(msg)
        from tiatoolbox_amd import _lib

        src = self._dev if self._dev.dim() == 3 else self._dev[..., None]  # noqa: PLR2004
        src = (src.to(torch.uint8) if src.dtype != torch.uint8 else src).contiguous()
        c = src.shape[-1]
        out = torch.empty((th, tw, c), dtype=torch.uint8, device=src.device)
        with torch.cuda.device(src.device):  # integer box sums on the device: no float32 copy of the slide
            rc = _lib.load().tia_box_downsample_u8(src.data_ptr(), h, w, c, factor, out.data_ptr(), _lib.current_stream())
        _lib.check(rc, "tia_box_downsample_u8")
        return out if self._dev.dim() == 3 else out[..., 0]  # noqa: PLR2004

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
