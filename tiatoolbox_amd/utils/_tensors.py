"""NumPy <-> device plumbing shared by the NumPy-facing API classes."""

from __future__ import annotations

import numpy as np
import torch

from tiatoolbox_amd import _lib


def default_device() -> torch.device:
    if not torch.cuda.is_available():
        msg = "No HIP device visible (torch.cuda.is_available() is False); there is no CPU fallback."
        raise _lib.HipLibraryError(msg)
    return torch.device("cuda", torch.cuda.current_device())


def to_device_batch(img, device: torch.device | None = None) -> tuple[torch.Tensor, str]:
    """Accept HWC / NHWC, NumPy / torch; return an NHWC uint8 CUDA batch and the input kind.

    kind is one of ``"np3"``, ``"np4"``, ``"t3"``, ``"t4"`` so results can be handed back
    in the caller's form.  Mirrors ``img.astype("uint8")`` of the reference entry points.
    """
    if isinstance(img, torch.Tensor):
        t = img
        kind = "t"
        if not t.is_cuda:
            t = t.to(device or default_device())
    else:
        arr = np.asarray(img)
        if arr.dtype != np.uint8:
            arr = arr.astype("uint8")
        t = torch.from_numpy(np.ascontiguousarray(arr)).to(device or default_device())
        kind = "np"
    if t.dtype != torch.uint8:
        t = t.to(torch.uint8)
    if t.dim() == 3:
        return t.unsqueeze(0).contiguous(), kind + "3"
    if t.dim() == 4:
        return t.contiguous(), kind + "4"
    msg = f"expected an HxWx3 image or NxHxWx3 batch, got shape {tuple(t.shape)}"
    raise ValueError(msg)


def from_device(out: torch.Tensor, kind: str):
    if kind.endswith("3"):
        out = out[0]
    if kind.startswith("np"):
        return out.cpu().numpy()
    return out
