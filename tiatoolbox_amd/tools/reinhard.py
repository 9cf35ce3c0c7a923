"""Reinhard colour normalisation (API of reference ``tiatoolbox/tools/stainnorm.py:222-367``).

The reference converts to 8-bit Lab, rescales each channel with float32 arithmetic, clips, truncates
to uint8 and converts back.  Because the Lab image is 8-bit, every per-pixel float operation is a
function of one byte: the whole chain becomes three 256-entry tables per image (evaluated with the
reference's own float32 arithmetic) between two fixed-point colour conversions.  GPU work: one
Lab-histogram kernel (statistics) and one fused RGB->Lab->LUT->RGB kernel.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from tiatoolbox_amd import _lib
from tiatoolbox_amd.tools.stainnorm import StainNormalizer
from tiatoolbox_amd.utils import _tensors, cvtables

_TABLES: dict[int, torch.Tensor] = {}


def lab_tables(device: torch.device) -> torch.Tensor:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _TABLES:
        fwd, inv = cvtables.lab_tables(), cvtables.lab_inverse_tables()
        host = _lib.LabTables()
        host.gamma[:] = fwd["gamma"].tolist()
        host.cbrt[:] = fwd["cbrt"].tolist()
        host.lab_y[:] = inv["y"].tolist()
        host.lab_ify[:] = inv["ify"].tolist()
        host.inv_gamma[:] = inv["inv_gamma"].tolist()
        host.c_fwd[:] = fwd["coeffs"].tolist()
        host.c_inv[:] = inv["coeffs"].tolist()
        raw = np.frombuffer(bytes(host), dtype=np.uint8).copy()
        _TABLES[idx] = torch.from_numpy(raw).to(torch.device("cuda", idx))
    return _TABLES[idx]


def lab_convert(src: torch.Tensor, direction: int) -> torch.Tensor:
    """8-bit ``RGB2LAB`` (0) / ``LAB2RGB`` (1) of a uint8 ``[...,3]`` CUDA tensor."""
    _lib.require_cuda(src)
    src = src.contiguous()
    out = torch.empty_like(src)
    with torch.cuda.device(src.device):
        rc = _lib.load().tia_lab_convert_u8(src.data_ptr(), src.numel() // 3, lab_tables(src.device).data_ptr(),
                                            direction, out.data_ptr(), _lib.current_stream())
    _lib.check(rc, "tia_lab_convert_u8")
    return out


def _channel_values() -> np.ndarray:
    """float32 value of each channel for every Lab byte after ``lab_split`` (:295-315)."""
    v = np.arange(256, dtype=np.float32)
    c1 = v.copy()
    c1 /= np.asarray(2.55)
    c2 = v.copy()
    c2 -= np.asarray(128.0)
    return np.stack([c1, c2, c2.copy()])


_CHAN_DEV: dict[int, torch.Tensor] = {}
_WORKSPACE: dict[int, torch.Tensor] = {}


def _workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """Per-device scratch of the fused transform (grown on demand, re-used: its lines stay in cache between calls)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ws = _WORKSPACE.get(idx)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=torch.device("cuda", idx))
        _WORKSPACE[idx] = ws
    return ws


def _channel_values_device(device: torch.device) -> torch.Tensor:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _CHAN_DEV:
        _CHAN_DEV[idx] = torch.from_numpy(np.ascontiguousarray(_channel_values())).to(torch.device("cuda", idx))
    return _CHAN_DEV[idx]


class ReinhardNormalizer(StainNormalizer):
    """Reinhard colour normaliser (ref. :222-367)."""

    def __init__(self) -> None:
        super().__init__()
        self.target_means: tuple[float, float, float]
        self.target_stds: tuple[float, float, float]

    # ---------------------------------------------------------------------------------- device
    @staticmethod
    def _lab_hist(batch: torch.Tensor) -> np.ndarray:  # (host copy of the Lab byte counts: fit / get_mean_std of one image)
        n, h, w, _ = batch.shape
        hist = torch.zeros((n, 3, 256), dtype=torch.int32, device=batch.device)
        with torch.cuda.device(batch.device):
            for s in range(0, n, 65535):
                m = min(65535, n - s)
                rc = _lib.load().tia_lab_hist_u8(batch[s:s + m].data_ptr(), m, h, w, lab_tables(batch.device).data_ptr(),
                                                 hist[s:s + m].data_ptr(), _lib.current_stream())
                _lib.check(rc, "tia_lab_hist_u8")
        return hist.cpu().numpy().astype(np.int64)

    @staticmethod
    def _mean_std(hist: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        """``cv2.meanStdDev`` of the float32 channels from exact byte counts: f64 mean, population std."""
        vals = _channel_values().astype(np.float64)[None]          # [1,3,256]
        n = hist.sum(-1, keepdims=True).astype(np.float64)
        mean = (hist * vals).sum(-1, keepdims=True) / n
        var = np.maximum((hist * vals * vals).sum(-1, keepdims=True) / n - mean * mean, 0.0)
        return mean[..., 0], np.sqrt(var)[..., 0]

    def _luts(self, means: np.ndarray, stds: np.ndarray) -> np.ndarray:
        """Per-image 3x256 uint8 tables: the reference's float32 chain (:283-292, 336-339) per Lab byte."""
        chan = _channel_values()                                  # float32 [3,256]
        if np.any(stds == 0):
            msg = "float division by zero"                        # the reference divides Python floats (:281-290)
            raise ZeroDivisionError(msg)
        # Python-float scalars meet float32 arrays: each scalar is rounded to float32, the ops are float32
        f32 = np.float32
        mean32 = means.astype(f32)[..., None]
        ratio32 = (np.asarray(self.target_stds, dtype=np.float64)[None] / stds).astype(f32)[..., None]
        tmean32 = np.asarray(self.target_means, dtype=np.float64).astype(f32)[None, :, None]
        norm = (chan[None] - mean32) * ratio32 + tmean32
        norm[:, 0] *= f32(2.55)
        norm[:, 1:] += f32(128.0)
        with np.errstate(invalid="ignore"):
            return np.clip(norm, 0, 255).astype(np.uint8)

    # ------------------------------------------------------------------------------------- API
    @staticmethod
    def lab_split(img):
        """uint8 RGB -> float32 ``L/2.55``, ``a-128``, ``b-128`` (ref. :295-315)."""
        batch, kind = _tensors.to_device_batch(img)
        lab = lab_convert(batch, 0).to(torch.float32)
        c1 = (lab[..., 0].to(torch.float64) / 2.55).to(torch.float32)
        c2, c3 = lab[..., 1] - 128.0, lab[..., 2] - 128.0
        return tuple(_tensors.from_device(c, kind) for c in (c1, c2, c3))

    @staticmethod
    def merge_back(chan1, chan2, chan3):
        """float32 Lab channels -> uint8 RGB (ref. :317-340)."""
        as_np = not isinstance(chan1, torch.Tensor)
        dev = _tensors.default_device()
        cs = [torch.as_tensor(np.asarray(c)).to(dev) if as_np else c for c in (chan1, chan2, chan3)]
        lab = torch.stack([cs[0] * np.float32(2.55), cs[1] + 128.0, cs[2] + 128.0], dim=-1)
        lab = torch.clamp(lab, 0, 255).to(torch.uint8)
        out = lab_convert(lab, 1)
        return out.cpu().numpy() if as_np else out

    def lab_statistics(self, batch: torch.Tensor) -> torch.Tensor:
        """Device-resident ``[n, 6]`` float64 (means of L/2.55, a-128, b-128, then the population stds) of a uint8 ``[n,h,w,3]``
        CUDA batch, no host round trip (what ``transform`` computes per image, :277-279): one launch (``tia_lab_moments_u8``) for
        patch shapes, the histogram kernel split over many workgroups per image + the moment kernel otherwise."""
        _lib.require_cuda(batch)
        batch = batch.contiguous()
        n, h, w, _ = batch.shape
        dev = batch.device
        meanstd = torch.empty((n, 6), dtype=torch.float64, device=dev)
        lib = _lib.load()
        tabs, chan = lab_tables(dev).data_ptr(), _channel_values_device(dev).data_ptr()
        with torch.cuda.device(dev):
            rc = lib.tia_lab_moments_u8(batch.data_ptr(), n, h, w, tabs, chan, meanstd.data_ptr(), 0, _lib.current_stream())
            if rc == 0:
                return meanstd
            if rc != _lib.TIA_ESIZE:
                _lib.check(rc, "tia_lab_moments_u8")
            hist = torch.zeros((n, 3, 256), dtype=torch.int32, device=dev)
            luts = torch.empty((n, 3, 256), dtype=torch.uint8, device=dev)
            one = (C.c_double * 3)(1.0, 1.0, 1.0)
            for s in range(0, n, 65535):
                m = min(65535, n - s)
                rc = lib.tia_lab_hist_u8(batch[s:s + m].data_ptr(), m, h, w, tabs, hist[s:s + m].data_ptr(), _lib.current_stream())
                _lib.check(rc, "tia_lab_hist_u8")
            rc = lib.tia_reinhard_luts(hist.data_ptr(), n, chan, one, one, luts.data_ptr(), meanstd.data_ptr(), 0, _lib.current_stream())
            _lib.check(rc, "tia_reinhard_luts")
        return meanstd

    def get_mean_std(self, img):
        batch, _ = _tensors.to_device_batch(img)
        mean, std = self._mean_std(self._lab_hist(batch))
        return tuple(float(v) for v in mean[0]), tuple(float(v) for v in std[0])

    def fit(self, target) -> None:
        self.target_means, self.target_stds = self.get_mean_std(target)

    def transform(self, img, *, out: str = "uint8"):
        """Per image: Lab statistics -> three byte tables -> RGB->Lab->table->RGB, no host round trip: one launch for patch
        shapes (``tia_reinhard_transform_u8``), three launches for large or ragged images.

        ``out``: ``"uint8"`` (reference behaviour, :342-367) or ``"unit_float16|bfloat16|float32"`` = ``ToTensor()`` of
        the uint8 result (what the engines feed the CNN; same keyword as :meth:`StainNormalizer.transform`).
        """
        unit = {"uint8": None, "unit_float16": torch.float16, "unit_bfloat16": torch.bfloat16,
                "unit_float32": torch.float32}
        if out not in unit:
            msg = f"ReinhardNormalizer.transform: unsupported out={out!r} (the Lab round trip is 8-bit)."
            raise ValueError(msg)
        unit_dtype = unit[out]
        batch, kind = _tensors.to_device_batch(img)
        n, h, w, _ = batch.shape
        dev = batch.device
        flags = torch.zeros(n, dtype=torch.int32, device=dev)
        out = torch.empty_like(batch)
        tmeans = (C.c_double * 3)(*[float(v) for v in self.target_means])
        tstds = (C.c_double * 3)(*[float(v) for v in self.target_stds])
        lib = _lib.load()
        tabs, chan = lab_tables(dev).data_ptr(), _channel_values_device(dev).data_ptr()
        with torch.cuda.device(dev):
            # patches: one launch (Lab kept in registers up to 256 x 256, in a cache-resident slot per workgroup above that)
            ws_bytes = int(lib.tia_reinhard_workspace_bytes(n, h, w))
            ws = _workspace(dev, ws_bytes) if ws_bytes else None
            rc = lib.tia_reinhard_transform_u8(batch.data_ptr(), n, h, w, tabs, chan, tmeans, tstds, out.data_ptr(), 0,
                                               flags.data_ptr(), ws.data_ptr() if ws is not None else 0, ws_bytes,
                                               _lib.current_stream())
            if rc not in (0, _lib.TIA_ESIZE):
                _lib.check(rc, "tia_reinhard_transform_u8")
            if rc == _lib.TIA_ESIZE:
                # large single images / ragged shapes: histogram over many workgroups per image -> tables -> apply
                hist = torch.zeros((n, 3, 256), dtype=torch.int32, device=dev)
                luts = torch.empty((n, 3, 256), dtype=torch.uint8, device=dev)
                for s in range(0, n, 65535):
                    m = min(65535, n - s)
                    rc = lib.tia_lab_hist_u8(batch[s:s + m].data_ptr(), m, h, w, tabs, hist[s:s + m].data_ptr(), _lib.current_stream())
                    _lib.check(rc, "tia_lab_hist_u8")
                rc = lib.tia_reinhard_luts(hist.data_ptr(), n, chan, tmeans, tstds, luts.data_ptr(), 0, flags.data_ptr(),
                                           _lib.current_stream())
                _lib.check(rc, "tia_reinhard_luts")
                for s in range(0, n, 65535):
                    m = min(65535, n - s)
                    rc = lib.tia_reinhard_apply_u8(batch[s:s + m].data_ptr(), m, h, w, tabs, luts[s:s + m].data_ptr(),
                                                   out[s:s + m].data_ptr(), _lib.current_stream())
                    _lib.check(rc, "tia_reinhard_apply_u8")
        if bool(flags.any()):
            msg = "float division by zero"  # the reference divides Python floats (stainnorm.py:281-290)
            raise ZeroDivisionError(msg)
        if unit_dtype is not None:
            out = out.to(torch.float32).div(255).to(unit_dtype)
        return _tensors.from_device(out, kind)
