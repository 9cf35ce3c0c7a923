"""``rgb2od`` / ``od2rgb`` (API of reference ``tiatoolbox/utils/transforms.py:209-256``).

Inside the kernels the OD conversion is a 256-entry table look-up fused with whatever
consumes it; these stand-alone functions exist for API compatibility.  Unlike the
reference, ``rgb2od`` does **not** write into its argument (the reference replaces zeros
by ones in place, ``transforms.py:229-230``).
"""

from __future__ import annotations

import numpy as np
import torch

from tiatoolbox_amd.utils import _tensors, cvtables


def rgb2od(img):
    """``max(-log(max(img,1)/255), 1e-6)`` as float64, same container kind as ``img``."""
    batch, kind = _tensors.to_device_batch(img)
    lut = torch.from_numpy(cvtables.od_lut()).to(batch.device)
    return _tensors.from_device(lut[batch.long()], kind)


def od2rgb(od: np.ndarray) -> np.ndarray:
    """``uint8(255*exp(-max(od,1e-6)))`` (truncation), e.g. for a (2,3) stain matrix."""
    od = np.maximum(od, 1e-6)
    return (255 * np.exp(-1 * od)).astype(np.uint8)
