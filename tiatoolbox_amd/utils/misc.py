"""Hot-path helpers with the reference's names (``tiatoolbox/utils/misc.py``).

Only the functions on the per-patch path are provided: ``contrast_enhancer``,
``get_luminosity_tissue_mask``, ``load_stain_matrix``, ``get_bounding_box``,
``cast_to_min_dtype``.  Image-sized work runs on the GPU through the C ABI.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from tiatoolbox_amd import _lib
from tiatoolbox_amd.tools import _stain_device as dev
from tiatoolbox_amd.utils import _tensors
from tiatoolbox_amd.utils.exceptions import FileNotSupportedError


def load_stain_matrix(stain_matrix_input) -> np.ndarray:
    """Reference ``utils/misc.py:218-258``."""
    if isinstance(stain_matrix_input, (str, Path)):
        suffix = Path(stain_matrix_input).suffix
        if suffix not in [".csv", ".npy"]:
            msg = "If supplying a path to a stain matrix, use either a npy or a csv file"
            raise FileNotSupportedError(msg)
        if suffix == ".csv":
            import pandas as pd

            return pd.read_csv(stain_matrix_input).to_numpy()
        return np.load(str(stain_matrix_input))
    if isinstance(stain_matrix_input, np.ndarray):
        return stain_matrix_input
    msg = "Stain_matrix must be either a path to npy/csv file or a numpy array"
    raise TypeError(msg)


def _mask_stats(batch: torch.Tensor, threshold: float) -> tuple[torch.Tensor, int]:
    params = dev.make_params(mode=_lib.MODE_FIXED, luminosity_threshold=threshold,
                             stain_fixed=np.array([[0.65, 0.70, 0.29], [0.07, 0.99, 0.11]]))
    return dev.stain_stats(batch, params), params.y_thr


def get_luminosity_tissue_mask(img, threshold: float):
    """Reference ``utils/misc.py:261-290``: contrast enhance -> 8-bit Lab L -> ``L/255 < thr``.

    Accepts an HWC image (NumPy/torch) or an NHWC batch; returns a bool mask of the same kind.
    Raises ``ValueError("Empty tissue mask computed.")`` like the reference.
    """
    batch, kind = _tensors.to_device_batch(img)
    stats, y_thr = _mask_stats(batch, threshold)
    mask = dev.luminosity_mask(batch, stats, y_thr)
    if not bool(mask.flatten(1).any(dim=1).all()):
        msg = "Empty tissue mask computed."
        raise ValueError(msg)
    return _tensors.from_device(mask, kind)


def contrast_enhancer(img: np.ndarray, low_p: int = 2, high_p: int = 98) -> np.ndarray:
    """Reference ``utils/misc.py:405-444``.

    The percentiles come from the GPU byte histogram (``tia_stain_stats_u8`` P1); the
    resulting 256-entry intensity map is applied as a table look-up on the device.
    """
    if not isinstance(img, torch.Tensor) and np.asarray(img).dtype != np.uint8:
        msg = "Image should be uint8."
        raise AssertionError(msg)
    batch, kind = _tensors.to_device_batch(img)
    params = dev.make_params(mode=_lib.MODE_FIXED,
                             stain_fixed=np.array([[0.65, 0.70, 0.29], [0.07, 0.99, 0.11]]))
    params.q_img_lo = float(np.true_divide(low_p, 100))
    params.q_img_hi = float(np.true_divide(high_p, 100))
    stats = dev.stain_stats(batch, params)
    plow = stats[:, _lib.ST_PLOW].view(-1, 1)
    phigh = stats[:, _lib.ST_PHIGH].view(-1, 1)
    v = torch.arange(256, dtype=torch.float64, device=batch.device).view(1, -1)
    x = torch.minimum(torch.maximum(v, plow), phigh)
    lut = torch.where(phigh > plow, (x - plow) / (phigh - plow) * 255.0 + 0.0, v).to(torch.uint8)
    n = batch.shape[0]
    batch = batch.contiguous()
    out = torch.empty_like(batch)
    lut = lut.contiguous()
    lib = _lib.load()
    with torch.cuda.device(batch.device):
        for s in range(0, n, 65535):  # one 256-byte table per image, applied by tia_lut_apply_u8 (no widened copy)
            m = min(65535, n - s)
            rc = lib.tia_lut_apply_u8(batch[s:s + m].data_ptr(), m, batch[0].numel(), lut[s:s + m].data_ptr(),
                                      out[s:s + m].data_ptr(), _lib.current_stream())
            _lib.check(rc, "tia_lut_apply_u8")
    return _tensors.from_device(out, kind)


def get_bounding_box(img: np.ndarray) -> np.ndarray:
    """Reference ``utils/misc.py:898-922``: ``[x_min, y_min, x_max+1, y_max+1]`` of non-zeros."""
    ys = np.flatnonzero(np.any(img, axis=1))
    xs = np.flatnonzero(np.any(img, axis=0))
    return np.array([xs[0], ys[0], xs[-1] + 1, ys[-1] + 1])


def cast_to_min_dtype(array):
    """Reference ``utils/misc.py:1925-1961``: bool if max is 1, else smallest unsigned int."""
    is_tensor = isinstance(array, torch.Tensor)
    max_value = int(array.max())
    if max_value == 1:
        return array.to(torch.bool) if is_tensor else array.astype(bool)
    candidates = ((np.uint8, torch.uint8), (np.uint16, torch.uint16), (np.uint32, torch.uint32),
                  (np.uint64, torch.uint64))
    for np_dt, t_dt in candidates:
        if max_value <= np.iinfo(np_dt).max:
            return array.to(t_dt) if is_tensor else array.astype(np_dt)
    return array
