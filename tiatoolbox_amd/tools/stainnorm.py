"""Stain normalisation (API of reference ``tiatoolbox/tools/stainnorm.py``).

``fit`` / ``transform`` keep the reference's signatures (HWC ``uint8`` NumPy in/out) and
additionally accept NHWC batches and ``torch`` CUDA tensors, which is what the engines use:
one ``tia_stain_stats_u8`` launch (a workgroup per patch) + one streaming
``tia_stain_apply_u8`` launch for the whole batch.
"""

from __future__ import annotations

import contextlib

import numpy as np
import torch

from tiatoolbox_amd import _lib
from tiatoolbox_amd.tools import _stain_device as dev
from tiatoolbox_amd.tools.stainextract import (
    CustomExtractor,
    MacenkoExtractor,
    RuifrokExtractor,
    VahadaneExtractor,
)
from tiatoolbox_amd.utils import _tensors
from tiatoolbox_amd.utils.exceptions import MethodNotSupportedError
from tiatoolbox_amd.utils.misc import load_stain_matrix
from tiatoolbox_amd.utils.transforms import od2rgb

_OUT_KINDS = {
    "uint8": _lib.OUT_U8, "float32": _lib.OUT_F32, "float64": _lib.OUT_F64,
    "unit_float16": _lib.OUT_UNIT_F16, "unit_bfloat16": _lib.OUT_UNIT_BF16, "unit_float32": _lib.OUT_UNIT_F32,
}


class StainNormalizer:
    """Stain normalisation base class (ref. :19-113).

    Attributes (as in the reference): ``extractor``, ``stain_matrix_target`` (2,3),
    ``target_concentrations`` (H*W,2), ``maxC_target`` (1,2), ``stain_matrix_target_RGB``.

    ``precision``: ``"f64"`` evaluates the per-pixel recomposition in float64 (fused stain matrix, table ``exp``: ulp-level
    equal to the reference, < 1e-12 on the 0..255 scale); ``"f64_ref"`` keeps the reference's order of operations with libm
    (the parity-audit mode); ``"f32"`` uses the fused 3x3 matrix in float32 with hardware ``exp2``
    (|error| <= 1e-4 on the pre-cast float, HBM-bound).  Per-patch statistics are always f64.
    """

    def __init__(self) -> None:
        self.extractor: CustomExtractor | MacenkoExtractor | RuifrokExtractor | VahadaneExtractor
        self.stain_matrix_target: np.ndarray
        self.maxC_target = None
        self.stain_matrix_target_RGB: np.ndarray
        self.precision = "f64"
        self._target_batch = None
        self._target_conc = None
        self._deferred: list | None = None

    @contextlib.contextmanager
    def deferred_checks(self):
        """Inside this scope ``transform`` does not synchronise to look at the per-patch error flags (empty tissue mask,
        degenerate statistics): they stay on the device and are raised ONCE, with the indices of the offending patches
        counted over the whole scope, when it ends normally (the engines wrap a run's loop in it)."""
        outer, self._deferred = self._deferred, []
        try:
            yield
            pending = self._deferred
        finally:
            self._deferred = outer
        if pending:
            dev.raise_on_flags(torch.cat(pending))

    # ------------------------------------------------------------------------------ helpers
    def _source_stats(self, batch: torch.Tensor, *, with_target: bool) -> torch.Tensor:
        kw = {}
        if with_target:
            kw = {"target_stain": self.stain_matrix_target, "target_maxc": self.maxC_target}
        if hasattr(self.extractor, "stats_params"):
            params = self.extractor.stats_params(**kw)
        else:  # extractor without a device implementation (Vahadane): per-image host solve
            if batch.shape[0] != 1:
                return torch.cat([self._source_stats(batch[i:i + 1], with_target=with_target)
                                  for i in range(batch.shape[0])])
            sm = self.extractor.get_stain_matrix(batch[0])
            params = dev.make_params(mode=_lib.MODE_FIXED, stain_fixed=sm, **kw)
        return dev.stain_stats(batch, params)

    @staticmethod
    def get_concentrations(img, stain_matrix: np.ndarray) -> np.ndarray:
        """Least-squares stain concentrations, (H*W, 2) float64 (ref. :49-66)."""
        batch, kind = _tensors.to_device_batch(img)
        params = dev.make_params(mode=_lib.MODE_FIXED, stain_fixed=stain_matrix)
        stats = dev.stain_stats(batch, params)
        conc = dev.concentrations(batch, stats)
        return _tensors.from_device(conc, kind)

    @property
    def target_concentrations(self) -> np.ndarray:
        """Concentration matrix of the fitted target (computed on first access)."""
        if self._target_conc is None:
            self._target_conc = self.get_concentrations(self._target_batch[0], self.stain_matrix_target)
            if isinstance(self._target_conc, torch.Tensor):
                self._target_conc = self._target_conc.cpu().numpy()
        return self._target_conc

    # ---------------------------------------------------------------------------------- API
    def fit(self, target) -> None:
        """Fit to a target image (ref. :68-87)."""
        batch, _ = _tensors.to_device_batch(target)
        if batch.shape[0] != 1:
            msg = "fit() expects a single HxWx3 target image."
            raise ValueError(msg)
        stats = self._source_stats(batch, with_target=False)
        dev.raise_on_flags(stats)
        host = stats[0].cpu().numpy()
        self.stain_matrix_target = host[_lib.ST_STAIN:_lib.ST_STAIN + 6].reshape(2, 3).copy()
        self.maxC_target = host[_lib.ST_MAXC:_lib.ST_MAXC + 2].reshape((1, 2)).copy()
        self.stain_matrix_target_RGB = od2rgb(self.stain_matrix_target)
        self._target_batch = batch
        self._target_conc = None

    def transform(self, img, *, out: str = "uint8", return_stats: bool = False):
        """Normalise an image or a batch (ref. :89-113).

        ``out``: ``"uint8"`` (reference behaviour), ``"float32"``/``"float64"`` (the value before
        the ``astype(uint8)`` truncation), or ``"unit_float16|bfloat16|float32"``
        (= ``ToTensor()`` of the uint8 result, ready for the CNN).
        """
        batch, kind = _tensors.to_device_batch(img)
        stats = self._source_stats(batch, with_target=True)
        math = {"f64": _lib.MATH_F64, "f64_ref": _lib.MATH_F64_REF}.get(self.precision, _lib.MATH_F32)
        res = dev.stain_apply(batch, stats, self.stain_matrix_target, out_kind=_OUT_KINDS[out], math=math)
        if self._deferred is not None:
            self._deferred.append(stats[:, _lib.ST_FLAGS].clone())
        else:
            dev.raise_on_flags(stats)
        res = _tensors.from_device(res, kind)
        return (res, stats) if return_stats else res


class CustomNormalizer(StainNormalizer):
    """User-defined stain matrix (ref. :116-141)."""

    def __init__(self, stain_matrix: np.ndarray) -> None:
        super().__init__()
        self.extractor = CustomExtractor(stain_matrix)


class RuifrokNormalizer(StainNormalizer):
    """Ruifrok & Johnston (ref. :144-166)."""

    def __init__(self) -> None:
        super().__init__()
        self.extractor = RuifrokExtractor()


class MacenkoNormalizer(StainNormalizer):
    """Macenko (ref. :169-192)."""

    def __init__(self) -> None:
        super().__init__()
        self.extractor = MacenkoExtractor()


class VahadaneNormalizer(StainNormalizer):
    """Vahadane (ref. :195-219)."""

    def __init__(self) -> None:
        super().__init__()
        self.extractor = VahadaneExtractor()


def get_normalizer(method_name: str, stain_matrix=None) -> StainNormalizer:
    """Factory (ref. :370-425)."""
    name = method_name.lower()
    if name not in ["reinhard", "ruifrok", "macenko", "vahadane", "custom"]:
        raise MethodNotSupportedError
    if stain_matrix is not None and name != "custom":
        msg = '`stain_matrix` is only defined when using `method_name`="custom".'
        raise ValueError(msg)
    if name == "reinhard":
        from tiatoolbox_amd.tools.reinhard import ReinhardNormalizer

        return ReinhardNormalizer()
    if name == "ruifrok":
        return RuifrokNormalizer()
    if name == "macenko":
        return MacenkoNormalizer()
    if name == "vahadane":
        return VahadaneNormalizer()
    if stain_matrix is None:
        msg = '`stain_matrix` is None when using `method_name`="custom".'
        raise ValueError(msg)
    return CustomNormalizer(load_stain_matrix(stain_matrix))
