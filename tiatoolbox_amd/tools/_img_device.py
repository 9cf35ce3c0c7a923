"""Device wrappers for the image primitives (grey, histogram, CCL, morphology, hole filling)."""

from __future__ import annotations

import numpy as np
import torch

from tiatoolbox_amd import _lib

_MAX_PLANES = 65535


def _call(name: str, *args) -> None:
    _lib.check(getattr(_lib.load(), name)(*args, _lib.current_stream()), name)


def rgb2gray(img: torch.Tensor) -> torch.Tensor:
    """uint8 [...,3] -> uint8 [...] (OpenCV 8-bit RGB2GRAY)."""
    _lib.require_cuda(img)
    img = img.contiguous()
    out = torch.empty(img.shape[:-1], dtype=torch.uint8, device=img.device)
    with torch.cuda.device(img.device):
        _call("tia_rgb2gray_u8", img.data_ptr(), out.numel(), out.data_ptr())
    return out


def hist256(data: torch.Tensor, hist: torch.Tensor | None = None) -> torch.Tensor:
    """Accumulate the byte histogram of a uint8 tensor into ``hist`` (int32 view of uint32[256])."""
    _lib.require_cuda(data)
    data = data.contiguous()
    if hist is None:
        hist = torch.zeros(256, dtype=torch.int32, device=data.device)
    with torch.cuda.device(data.device):
        _call("tia_hist256_u8", data.data_ptr(), data.numel(), hist.data_ptr())
    return hist


def threshold_lt(src: torch.Tensor, thr: int, *, is_rgb: bool) -> torch.Tensor:
    _lib.require_cuda(src)
    src = src.contiguous()
    shape = src.shape[:-1] if is_rgb else src.shape
    out = torch.empty(shape, dtype=torch.uint8, device=src.device)
    with torch.cuda.device(src.device):
        _call("tia_threshold_lt_u8", src.data_ptr(), out.numel(), int(is_rgb), int(thr), out.data_ptr())
    return out


def _planes(mask: torch.Tensor) -> tuple[torch.Tensor, int, int, int]:
    _lib.require_cuda(mask)
    if mask.dim() == 2:
        mask = mask.unsqueeze(0)
    if mask.dim() != 3:
        msg = f"expected [n,h,w] planes, got {tuple(mask.shape)}"
        raise ValueError(msg)
    n, h, w = mask.shape
    if n > _MAX_PLANES:
        msg = "at most 65535 planes per call"
        raise ValueError(msg)
    return mask.contiguous(), n, h, w


def ccl_label(mask: torch.Tensor, connectivity: int = 4) -> tuple[torch.Tensor, torch.Tensor]:
    """Labels (int32, 1..K in raster order of first pixel) and per-plane counts."""
    m, n, h, w = _planes(mask.to(torch.uint8) if mask.dtype != torch.uint8 else mask)
    labels = torch.empty((n, h, w), dtype=torch.int32, device=m.device)
    count = torch.empty(n, dtype=torch.int32, device=m.device)
    ws = torch.empty(n * h * w, dtype=torch.int32, device=m.device)
    with torch.cuda.device(m.device):
        _call("tia_ccl_label_i32", m.data_ptr(), n, h, w, connectivity, labels.data_ptr(), count.data_ptr(),
              ws.data_ptr())
    return labels, count


def label_area_filter(labels: torch.Tensor, min_keep: int) -> torch.Tensor:
    """In place: zero labels whose area is < ``min_keep`` (no relabelling)."""
    _lib.require_cuda(labels)
    n, h, w = labels.shape
    ws = torch.empty(n * (h * w + 1), dtype=torch.int32, device=labels.device)
    with torch.cuda.device(labels.device):
        _call("tia_label_area_filter_i32", labels.data_ptr(), n, h, w, int(min_keep), ws.data_ptr())
    return labels


def offsets_of(kernel: np.ndarray, device: torch.device) -> torch.Tensor:
    """(dy, dx) offsets of the non-zero entries of a structuring element, anchor at the centre."""
    kh, kw = kernel.shape
    ys, xs = np.nonzero(kernel)
    offs = np.stack([ys - kh // 2, xs - kw // 2], axis=1).astype(np.int32)
    return torch.from_numpy(np.ascontiguousarray(offs)).to(device)


def binary_morph(mask: torch.Tensor, offsets: torch.Tensor, op: str) -> torch.Tensor:
    m, n, h, w = _planes(mask)
    out = torch.empty_like(m)
    with torch.cuda.device(m.device):
        _call("tia_binary_morph_u8", m.data_ptr(), n, h, w, offsets.data_ptr(), offsets.shape[0],
              {"dilate": 0, "erode": 1}[op], out.data_ptr())
    return out


def fill_holes(mask: torch.Tensor) -> torch.Tensor:
    m, n, h, w = _planes(mask)
    out = torch.empty_like(m)
    ws = torch.empty(2 * n * h * w + n, dtype=torch.int32, device=m.device)
    with torch.cuda.device(m.device):
        _call("tia_fill_holes_u8", m.data_ptr(), n, h, w, out.data_ptr(), ws.data_ptr())
    return out


def get_structuring_element_ellipse(ksize: tuple[int, int]) -> np.ndarray:
    """``cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (w, h))`` (tiny, host side)."""
    w, h = int(ksize[0]), int(ksize[1])
    elem = np.zeros((h, w), dtype=np.uint8)
    if (w, h) == (1, 1):
        elem[:] = 1
        return elem
    r, c = h // 2, w // 2
    inv_r2 = 1.0 / (r * r) if r else 0.0
    for i in range(h):
        dy = i - r
        if abs(dy) <= r:
            dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * inv_r2)))
            elem[i, max(c - dx, 0):min(c + dx + 1, w)] = 1
    return elem
