"""Small tensor helpers (API of reference ``tiatoolbox/models/architecture/utils.py``)."""

from __future__ import annotations

import numpy as np
import torch


def argmax_last_axis(image):
    """``argmax`` over the last axis (ref. :391-405); NumPy or torch."""
    if isinstance(image, torch.Tensor):
        return torch.argmax(image, dim=-1)
    return np.argmax(image, axis=-1)


def centre_crop(img, crop_shape, data_format: str = "NCHW"):
    """Crop ``crop_shape`` = (h, w) pixels in total, split evenly top/bottom, left/right (ref. :54-111)."""
    if data_format not in ["NCHW", "NHWC"]:
        msg = f"Unknown input format `{data_format}`."
        raise ValueError(msg)
    crop_t = crop_shape[0] // 2
    crop_b = crop_shape[0] - crop_t
    crop_l = crop_shape[1] // 2
    crop_r = crop_shape[1] - crop_l
    if data_format == "NCHW":
        return img[:, :, crop_t:img.shape[2] - crop_b, crop_l:img.shape[3] - crop_r]
    return img[:, crop_t:img.shape[1] - crop_b, crop_l:img.shape[2] - crop_r, :]
