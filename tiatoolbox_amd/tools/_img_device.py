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


def gray_hist(img: torch.Tensor, *, channels: int, hist: torch.Tensor | None = None) -> torch.Tensor:
    """Grey conversion fused with the 256-bin histogram: uint8 RGB pixels (``channels=3``) or a grey plane (``channels=1``) ->
    counts accumulated into ``hist`` (int32[256]); one pass, nothing written but the counts."""
    _lib.require_cuda(img)
    img = img.contiguous()
    if hist is None:
        hist = torch.zeros(256, dtype=torch.int32, device=img.device)
    with torch.cuda.device(img.device):
        _call("tia_gray_hist_u8", img.data_ptr(), img.numel() // channels, channels, hist.data_ptr())
    return hist


def otsu_fit(img: torch.Tensor, *, channels: int) -> torch.Tensor:
    """``OtsuTissueMasker.fit`` of uint8 pixels in one launch (grey + histogram; the last workgroup runs Otsu's arithmetic): int32[2] =
    (threshold, occupied bins) on the device.  One small buffer carries counts, ticket and result: one fill, one launch."""
    _lib.require_cuda(img)
    img = img.contiguous()
    buf = torch.zeros(260, dtype=torch.int32, device=img.device)  # [0:256] counts, [256] ticket, [258:260] result
    with torch.cuda.device(img.device):
        _call("tia_otsu_fit_u8", img.data_ptr(), img.numel() // channels, channels, buf.data_ptr(), buf[258:].data_ptr())
    return buf[258:260]


def otsu_threshold(hist: torch.Tensor) -> torch.Tensor:
    """``skimage.filters.threshold_otsu`` of byte counts on the device: int32[2] = (threshold, occupied bins)."""
    _lib.require_cuda(hist)
    out = torch.empty(2, dtype=torch.int32, device=hist.device)
    with torch.cuda.device(hist.device):
        _call("tia_otsu_threshold_u32", hist.data_ptr(), out.data_ptr())
    return out


def threshold_lt(src: torch.Tensor, thr: int | torch.Tensor, *, is_rgb: bool, dtype: torch.dtype = torch.uint8) -> torch.Tensor:
    """``grey < thr`` as 0/1 bytes (``dtype`` uint8, or bool: same storage, no conversion pass); ``thr`` is a Python integer or a
    device int32 tensor (read by the kernel: no host round trip)."""
    _lib.require_cuda(src)
    src = src.contiguous()
    shape = src.shape[:-1] if is_rgb else src.shape
    out = torch.empty(shape, dtype=dtype, device=src.device)
    with torch.cuda.device(src.device):
        if isinstance(thr, torch.Tensor):
            _call("tia_threshold_lt_dev_u8", src.data_ptr(), out.numel(), int(is_rgb), thr.data_ptr(), out.data_ptr())
        else:
            _call("tia_threshold_lt_u8", src.data_ptr(), out.numel(), int(is_rgb), int(thr), out.data_ptr())
    return out


def morph_mask(images: torch.Tensor, thr: int | torch.Tensor, min_region: int, offsets: torch.Tensor, *,
               channels: int, reach: int | None = None) -> torch.Tensor | None:
    """``MorphologicalMasker.transform`` in one launch: ``dilate(remove_small_objects(grey < thr, min_region, 8), element)`` of a
    uint8 ``[n,h,w,3]`` (RGB) or ``[n,h,w]`` (grey) batch as a bool ``[n,h,w]`` tensor, or ``None`` when the element's reach +
    ``min_region`` - 1 exceeds the 40-pixel halo of the tile kernel (the caller takes the multi-launch form)."""
    _lib.require_cuda(images)
    images = images.contiguous()
    n, h, w = images.shape[:3]
    if n > _MAX_PLANES:
        return None
    out = torch.empty((n, h, w), dtype=torch.bool, device=images.device)
    if reach is None:
        reach = int(offsets.abs().max().item()) if offsets.numel() else 0
    dev_thr = isinstance(thr, torch.Tensor)
    with torch.cuda.device(images.device):
        rc = _lib.load().tia_morph_mask_u8(images.data_ptr(), n, h, w, channels, 0 if dev_thr else int(thr),
                                           thr.data_ptr() if dev_thr else 0, int(min_region), offsets.data_ptr(), offsets.shape[0],
                                           reach, out.data_ptr(), _lib.current_stream())
    if rc == _lib.TIA_ESIZE:
        return None
    _lib.check(rc, "tia_morph_mask_u8")
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
