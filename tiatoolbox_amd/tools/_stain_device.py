"""Batched device entry points for the stain path (thin wrappers over the C ABI).

Everything here takes/returns ``torch`` CUDA tensors (NHWC uint8 patch batches) and
enqueues on the current HIP stream without synchronising.  The NumPy-facing classes in
``stainextract.py`` / ``stainnorm.py`` / ``stainaugment.py`` are built on these.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from tiatoolbox_amd import _lib
from tiatoolbox_amd.utils import cvtables

_TABLES: dict[int, torch.Tensor] = {}
_MAX_GRID_Y = 65535


def tables(device: torch.device) -> torch.Tensor:
    """Device copy of ``tia_stain_tables`` (uploaded once per device)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _TABLES:
        host = _lib.StainTables()
        od = cvtables.od_lut()
        C.memmove(host.od_lut, od.ctypes.data, od.nbytes)
        od32 = od.astype(np.float32)
        C.memmove(host.od_lut_f32, od32.ctypes.data, od32.nbytes)
        ty = np.ascontiguousarray(cvtables.ty_tables())
        C.memmove(host.ty, ty.ctypes.data, ty.nbytes)
        raw = np.frombuffer(bytes(host), dtype=np.uint8).copy()
        _TABLES[idx] = torch.from_numpy(raw).to(torch.device("cuda", idx))
    return _TABLES[idx]


_WS: dict[int, torch.Tensor] = {}


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Grow-only scratch per device (kernels on one stream are ordered, so it can be shared)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    cur = _WS.get(idx)
    if cur is None or cur.numel() < nbytes:
        cur = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=torch.device("cuda", idx))
        _WS[idx] = cur
    return cur


def as_batch(img: torch.Tensor) -> torch.Tensor:
    """Validate an NHWC uint8 CUDA batch (contiguous)."""
    _lib.require_cuda(img, "image batch")
    if img.dtype != torch.uint8 or img.dim() != 4 or img.shape[-1] != 3:
        msg = f"expected a uint8 NHWC batch with 3 channels, got {tuple(img.shape)} {img.dtype}"
        raise ValueError(msg)
    return img.contiguous()


def make_params(*, mode: int, luminosity_threshold: float = 0.8, angular_percentile: float = 99,
                stain_fixed: np.ndarray | None = None, target_stain: np.ndarray | None = None,
                target_maxc: np.ndarray | None = None, zero_to_one: bool = False, dl_alpha: float = 0.1,
                dl_tol: float = 1e-8, dl_max_iter: int = 3, dl_seed: int = 0, select_mode: int = 0,
                dl_one_kernel: bool = False) -> _lib.StainParams:
    p = _lib.StainParams()
    p.select_mode = int(select_mode)
    p.dl_one_kernel = int(bool(dl_one_kernel))
    p.dl_alpha, p.dl_tol, p.dl_max_iter, p.dl_seed = float(dl_alpha), float(dl_tol), int(dl_max_iter), int(dl_seed)
    # np.percentile divides q by 100 in float64 (numpy/lib/_function_base_impl.py: percentile)
    p.q_img_lo = float(np.true_divide(2, 100))
    p.q_img_hi = float(np.true_divide(98, 100))
    p.q_phi_lo = float(np.true_divide(100 - angular_percentile, 100))
    p.q_phi_hi = float(np.true_divide(angular_percentile, 100))
    p.q_conc = float(np.true_divide(99, 100))
    p.y_thr = cvtables.y_threshold(luminosity_threshold)
    p.mode = mode
    p.zero_to_one = int(zero_to_one)
    if stain_fixed is not None:
        sf = np.asarray(stain_fixed, dtype=np.float64)
        if sf.shape != (2, 3):
            msg = "The batched normaliser needs a (2, 3) stain matrix."
            raise ValueError(msg)
        p.stain_fixed[:] = sf.ravel().tolist()
    if target_stain is not None:
        p.has_target = 1
        p.target_stain[:] = np.asarray(target_stain, dtype=np.float64).ravel().tolist()
        p.target_maxc[:] = np.asarray(target_maxc, dtype=np.float64).ravel().tolist()
    return p


_VAHADANE_CHUNK_BYTES = 8 << 30  # dictionary scratch per launch (16 B per pixel per patch): bigger batches go in chunks


def stain_stats(img: torch.Tensor, params: _lib.StainParams, stain_given: torch.Tensor | None = None) -> torch.Tensor:
    """Per-patch statistics ``[N, 64]`` float64 (see ``include/tiatoolbox_amd.h``).  ``MODE_GIVEN``: ``stain_given``
    ``[N, 2, 3]`` float64 = each patch's own stain matrix."""
    img = as_batch(img)
    n, h, w, _ = img.shape
    stats = torch.empty((n, _lib.TIA_STATS_STRIDE), dtype=torch.float64, device=img.device)
    if params.mode == _lib.MODE_GIVEN:
        if stain_given is None or tuple(stain_given.shape) != (n, 2, 3):
            msg = "MODE_GIVEN needs an [N, 2, 3] stain matrix tensor."
            raise ValueError(msg)
        stats[:, _lib.ST_STAIN:_lib.ST_STAIN + 6] = stain_given.to(device=img.device, dtype=torch.float64).reshape(n, 6)
    tab = tables(img.device)
    lib = _lib.load()
    chunk = n
    if params.mode == _lib.MODE_VAHADANE:
        chunk = max(1, min(n, _VAHADANE_CHUNK_BYTES // (h * w * 20)))
    ws_bytes = lib.tia_stain_stats_workspace_bytes_mode(chunk, h, w, params.mode)
    ws = _workspace(img.device, ws_bytes)
    with torch.cuda.device(img.device):
        for s in range(0, n, chunk):
            m = min(chunk, n - s)
            rc = lib.tia_stain_stats_u8(img[s:s + m].data_ptr(), m, h, w, tab.data_ptr(), C.byref(params),
                                        stats[s:s + m].data_ptr(), ws.data_ptr(), ws_bytes, _lib.current_stream())
            _lib.check(rc, "tia_stain_stats_u8")
    return stats


def redo_count(device: torch.device, n: int, h: int, w: int) -> int:
    """Diagnostics: how many patches of the LAST ``stain_stats`` launch of ``n`` patches of ``h x w`` on ``device`` the
    register-resident kernel handed back to the streaming kernel (the flag array sits behind the bin cache in the workspace)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ws = _WS.get(idx)
    off = (n * h * w * 4 + 255) & ~255
    if ws is None or ws.numel() < off + 4 * n:
        return -1
    return int(ws[off:off + 4 * n].view(torch.int32).ne(0).sum().item())


_OUT_DTYPES = {
    _lib.OUT_U8: torch.uint8, _lib.OUT_F32: torch.float32, _lib.OUT_F64: torch.float64,
    _lib.OUT_UNIT_F16: torch.float16, _lib.OUT_UNIT_BF16: torch.bfloat16, _lib.OUT_UNIT_F32: torch.float32,
}


def stain_apply(img: torch.Tensor, stats: torch.Tensor, target_stain: np.ndarray, *,
                out_kind: int = _lib.OUT_U8, math: int = _lib.MATH_F64,
                out: torch.Tensor | None = None) -> torch.Tensor:
    img = as_batch(img)
    n, h, w, _ = img.shape
    if out is None:
        out = torch.empty((n, h, w, 3), dtype=_OUT_DTYPES[out_kind], device=img.device)
    tab = tables(img.device)
    ts = (C.c_double * 6)(*np.asarray(target_stain, dtype=np.float64).ravel().tolist())
    lib = _lib.load()
    with torch.cuda.device(img.device):
        for s in range(0, n, _MAX_GRID_Y):
            m = min(_MAX_GRID_Y, n - s)
            rc = lib.tia_stain_apply_u8(img[s:s + m].data_ptr(), m, h, w, tab.data_ptr(),
                                        stats[s:s + m].data_ptr(), ts, out[s:s + m].data_ptr(),
                                        out_kind, math, _lib.current_stream())
            _lib.check(rc, "tia_stain_apply_u8")
    return out


def concentrations(img: torch.Tensor, stats: torch.Tensor) -> torch.Tensor:
    img = as_batch(img)
    n, h, w, _ = img.shape
    out = torch.empty((n, h * w, 2), dtype=torch.float64, device=img.device)
    tab = tables(img.device)
    lib = _lib.load()
    with torch.cuda.device(img.device):
        for s in range(0, n, _MAX_GRID_Y):
            m = min(_MAX_GRID_Y, n - s)
            rc = lib.tia_stain_concentrations_f64(img[s:s + m].data_ptr(), m, h, w, tab.data_ptr(),
                                                  stats[s:s + m].data_ptr(), out[s:s + m].data_ptr(),
                                                  _lib.current_stream())
            _lib.check(rc, "tia_stain_concentrations_f64")
    return out


def luminosity_mask(img: torch.Tensor, stats: torch.Tensor, y_thr: int, *, zero_to_one: bool = False) -> torch.Tensor:
    img = as_batch(img)
    n, h, w, _ = img.shape
    out = torch.empty((n, h, w), dtype=torch.bool, device=img.device)  # the kernel writes 0 / 1 bytes: a bool tensor's storage
    tab = tables(img.device)
    lib = _lib.load()
    with torch.cuda.device(img.device):
        for s in range(0, n, _MAX_GRID_Y):
            m = min(_MAX_GRID_Y, n - s)
            rc = lib.tia_luminosity_mask_u8(img[s:s + m].data_ptr(), m, h, w, tab.data_ptr(),
                                            stats[s:s + m].data_ptr(), y_thr, int(zero_to_one),
                                            out[s:s + m].data_ptr(), _lib.current_stream())
            _lib.check(rc, "tia_luminosity_mask_u8")
    return out


def augment(img: torch.Tensor, stats: torch.Tensor, alpha_beta: torch.Tensor, y_thr: int, *,
            augment_background: bool, zero_to_one: bool, math: int = _lib.MATH_F64) -> torch.Tensor:
    img = as_batch(img)
    n, h, w, _ = img.shape
    _lib.require_cuda(alpha_beta, "alpha_beta")
    ab = alpha_beta.to(torch.float64).contiguous()
    out = torch.empty_like(img)
    tab = tables(img.device)
    lib = _lib.load()
    with torch.cuda.device(img.device):
        for s in range(0, n, _MAX_GRID_Y):
            m = min(_MAX_GRID_Y, n - s)
            rc = lib.tia_stain_augment_u8(img[s:s + m].data_ptr(), m, h, w, tab.data_ptr(),
                                          stats[s:s + m].data_ptr(), ab[s:s + m].data_ptr(), y_thr,
                                          int(augment_background), int(zero_to_one),
                                          out[s:s + m].data_ptr(), math, _lib.current_stream())
            _lib.check(rc, "tia_stain_augment_u8")
    return out


def raise_on_flags(stats: torch.Tensor) -> None:
    """Data-dependent errors detected on the device, raised like the reference does.  ``stats``: the ``[N, 64]`` statistics or
    just their flag column ``[N]``."""
    flags = (stats[:, _lib.ST_FLAGS] if stats.dim() == 2 else stats).to(torch.int64)
    empty = torch.nonzero(flags & _lib.FLAG_EMPTY_MASK).flatten()
    if empty.numel():
        msg = "Empty tissue mask computed."
        if stats.shape[0] > 1:
            msg += f" (patch indices {empty.tolist()[:16]})"
        raise ValueError(msg)  # utils/misc.py:286-288
    degenerate = torch.nonzero(flags & _lib.FLAG_DEGENERATE).flatten()
    if degenerate.numel():
        # fewer than two tissue pixels (np.cov with ddof=1 is NaN and np.linalg.eigh does not converge in the
        # reference, stainextract.py:202-205) or non-finite statistics (e.g. a zero 99th-percentile concentration,
        # stainnorm.py:103-104): never let them flow silently into the apply kernel
        msg = "Degenerate stain statistics (fewer than two tissue pixels, or non-finite stain matrix / maxC)."
        if stats.shape[0] > 1:
            msg += f" (patch indices {degenerate.tolist()[:16]})"
        raise np.linalg.LinAlgError(msg)
