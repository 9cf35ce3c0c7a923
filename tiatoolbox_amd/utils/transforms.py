"""``rgb2od`` / ``od2rgb`` (API of reference ``tiatoolbox/utils/transforms.py:209-256``).

Inside the stain kernels the OD conversion is a 256-entry table look-up fused with whatever
consumes it; the stand-alone ``rgb2od`` is ``tia_rgb2od_u8`` (same table, one streaming launch).
Like the reference (``transforms.py:229-230``) it replaces zeros by ones **in its argument**:
a writable NumPy array or a CUDA tensor is edited in place (``mutate=False`` switches that off).
"""

from __future__ import annotations

import numpy as np
import torch

from tiatoolbox_amd import _lib
from tiatoolbox_amd.utils import _tensors


def rgb2od(img, *, mutate: bool = True):
    """``max(-log(max(img,1)/255), 1e-6)`` as float64, any shape, same container kind as ``img``.

    Side effect as in the reference: ``img[img == 0] = 1`` on a writable uint8 NumPy array (applied on the HOST -- the bytes
    travel to the device once and nothing comes back but the result) and on a uint8 tensor (on the device, by the kernel).
    The kernel reads bytes: an input of another dtype is a ``TypeError`` (the reference would take the logarithm of float values
    as they are; cast to uint8 explicitly, as every call on the hot path does).
    """
    from tiatoolbox_amd.tools import _stain_device as dev

    is_tensor = isinstance(img, torch.Tensor)
    dtype_ok = (img.dtype == torch.uint8) if is_tensor else (np.asarray(img).dtype == np.uint8)
    if not dtype_ok:
        msg = f"rgb2od takes uint8 images on the device path, got {img.dtype if hasattr(img, 'dtype') else type(img).__name__}."
        raise TypeError(msg)
    kernel_mutates = False
    if is_tensor:
        src = img if img.is_cuda else img.to(_tensors.default_device())
        same_storage = src is img and img.is_contiguous()
        dev_img = src.contiguous()
        kernel_mutates = bool(mutate)
    else:
        arr = np.asarray(img)
        same_storage = False
        if mutate and isinstance(img, np.ndarray) and img.flags.writeable:
            img[img == 0] = 1  # the reference's side effect, on the caller's array; the kernel then has nothing to edit
        host = np.ascontiguousarray(arr)
        if not host.flags.writeable:  # torch refuses to wrap read-only memory silently
            host = host.copy()
        dev_img = torch.from_numpy(host).to(_tensors.default_device())
    out = torch.empty(dev_img.shape, dtype=torch.float64, device=dev_img.device)
    if dev_img.numel():
        lib = _lib.load()
        with torch.cuda.device(dev_img.device):
            rc = lib.tia_rgb2od_u8(dev_img.data_ptr(), dev_img.numel(), dev.tables(dev_img.device).data_ptr(),
                                   int(kernel_mutates), out.data_ptr(), _lib.current_stream())
        _lib.check(rc, "tia_rgb2od_u8")
    if kernel_mutates and not same_storage:
        img.copy_(dev_img)          # host tensor / non-contiguous view: hand the edit back
    return out if is_tensor else out.cpu().numpy()


def od2rgb(od: np.ndarray) -> np.ndarray:
    """``uint8(255*exp(-max(od,1e-6)))`` (truncation), e.g. for a (2,3) stain matrix."""
    od = np.maximum(od, 1e-6)
    return (255 * np.exp(-1 * od)).astype(np.uint8)
