"""Pre-processing registry (API of reference ``tiatoolbox/models/dataset/classification.py``)."""

from __future__ import annotations

import numpy as np
import torch


class UnitUInt8:
    """A uint8 NHWC device batch whose ``ToTensor`` step (``float32 / 255``) has been DEFERRED into the consumer: the
    hand-written ResNet stem divides by 255 while it loads the bytes, so the 4x larger float32 copy of the batch is never
    written.  The engines unwrap it and call the model directly (``EngineABC._forward_batch``)."""

    __slots__ = ("data",)

    def __init__(self, data: torch.Tensor) -> None:
        if data.dtype != torch.uint8:
            msg = "UnitUInt8 wraps a uint8 batch."
            raise TypeError(msg)
        self.data = data

    def __len__(self) -> int:
        return self.data.shape[0]


class _TorchPreprocCaller:
    """``ToTensor()`` followed by ``permute(1, 2, 0)``: uint8 HWC -> float32 HWC in [0, 1] (ref. :15-32).

    ``device_batch`` is the batched on-device form the engines use; with ``defer_unit_scale=True`` a uint8 batch comes
    back wrapped in :class:`UnitUInt8` (same values, the division happens in the model's stem kernel).
    """

    supports_deferred_unit_scale = True

    def __init__(self, preprocs: list) -> None:
        self.preprocs = preprocs

    def __call__(self, img) -> torch.Tensor:
        arr = np.asarray(img)
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if t.dtype == torch.uint8:
            return t.to(torch.float32).div(255)
        return t.to(torch.float32) if t.dtype != torch.float32 else t

    @staticmethod
    def device_batch(batch: torch.Tensor, dtype: torch.dtype, *, defer_unit_scale: bool = False):
        if batch.dtype == torch.uint8:
            if defer_unit_scale and batch.is_cuda:
                return UnitUInt8(batch.contiguous())
            return batch.to(torch.float32).div(255).to(dtype)
        return batch.to(dtype)


def predefined_preproc_func(dataset_name: str) -> _TorchPreprocCaller:
    """Pre-processing used by the pretrained models of a dataset (ref. :35-63)."""
    preproc_dict = {"kather100k": ["ToTensor"], "pcam": ["ToTensor"]}
    if dataset_name not in preproc_dict:
        msg = f"Predefined preprocessing for dataset `{dataset_name}` does not exist."
        raise ValueError(msg)
    return _TorchPreprocCaller(preproc_dict[dataset_name])


class StainNormPreproc:
    """Stain-normalise, then apply the model's own pre-processing.

    The reference injects stain normalisation by *replacing* ``model.preproc_func``
    (``models_abc.py:168-171``), which drops the dataset's ``ToTensor`` step unless the user
    composes both.  This callable is that composition: picklable for DataLoader workers,
    and recognised by the engines, which run it batched on the GPU
    (``tia_stain_stats_u8`` + ``tia_stain_apply_u8`` writing the CNN input directly).
    """

    def __init__(self, normalizer, then: _TorchPreprocCaller | None = None) -> None:
        self.normalizer = normalizer
        self.then = then if then is not None else predefined_preproc_func("kather100k")

    def __call__(self, img):
        return self.then(self.normalizer.transform(img))

    @property
    def supports_deferred_unit_scale(self) -> bool:
        return isinstance(self.then, _TorchPreprocCaller)

    def device_batch(self, batch: torch.Tensor, dtype: torch.dtype, *, defer_unit_scale: bool = False):
        if defer_unit_scale and self.supports_deferred_unit_scale and batch.is_cuda:
            return UnitUInt8(self.normalizer.transform(batch, out="uint8"))  # the reference's uint8 result, 1 B / value
        kind = {torch.float16: "unit_float16", torch.bfloat16: "unit_bfloat16",
                torch.float32: "unit_float32"}[dtype]
        return self.normalizer.transform(batch, out=kind)
