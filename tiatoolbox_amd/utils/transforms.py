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
    The kernel reads bytes.  Any other dtype (Python lists and integer arrays, float RGB) keeps the reference's behaviour --
    ``img[img == 0] = 1`` in place, then ``max(-log(img / 255), 1e-6)`` in the array's own floating type (float64 for integers) --
    evaluated with torch on the device: an API corner, not the hot path (every call there passes uint8).
    """
    from tiatoolbox_amd.tools import _stain_device as dev

    is_tensor = isinstance(img, torch.Tensor)
    dtype_ok = (img.dtype == torch.uint8) if is_tensor else (np.asarray(img).dtype == np.uint8)
    if not dtype_ok:
        return _rgb2od_any_dtype(img, mutate=mutate)
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


def _rgb2od_any_dtype(img, *, mutate: bool):
    """Reference ``utils/transforms.py:229-231`` for non-uint8 input (see :func:`rgb2od`)."""
    is_tensor = isinstance(img, torch.Tensor)
    if is_tensor:
        if mutate:
            img[img == 0] = 1
        t = img if img.is_cuda else img.to(_tensors.default_device())
    else:
        arr = img if isinstance(img, np.ndarray) else np.asarray(img)
        if arr.dtype == bool or arr.dtype.kind not in "iuf":
            msg = f"rgb2od takes numeric images, got {arr.dtype}."
            raise TypeError(msg)
        if mutate and isinstance(img, np.ndarray) and img.flags.writeable:
            img[img == 0] = 1
        else:
            arr = np.where(arr == 0, 1, arr)
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(_tensors.default_device())
    out_dtype = t.dtype if t.is_floating_point() else torch.float64  # the reference computes in the array's own floating type
    work = t.to(torch.float64)
    if is_tensor and not mutate:
        work = torch.where(work == 0, torch.ones_like(work), work)
    out = torch.clamp_min(-torch.log(work / 255), 1e-6).to(out_dtype)  # float64 on the device, rounded once to the result type
    return out if is_tensor else out.cpu().numpy()


def od2rgb(od: np.ndarray) -> np.ndarray:
    """``uint8(255*exp(-max(od,1e-6)))`` (truncation), e.g. for a (2,3) stain matrix."""
    od = np.maximum(od, 1e-6)
    return (255 * np.exp(-1 * od)).astype(np.uint8)
