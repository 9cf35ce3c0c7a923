"""Pre-processing registry (API of reference ``tiatoolbox/models/dataset/classification.py``)."""

from __future__ import annotations

import numpy as np
import torch


class _TorchPreprocCaller:
    """``ToTensor()`` followed by ``permute(1, 2, 0)``: uint8 HWC -> float32 HWC in [0, 1] (ref. :15-32).

    ``device_batch`` is the batched on-device form the engines use.
    """

    def __init__(self, preprocs: list) -> None:
        self.preprocs = preprocs

    def __call__(self, img) -> torch.Tensor:
        arr = np.asarray(img)
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if t.dtype == torch.uint8:
            return t.to(torch.float32).div(255)
        return t.to(torch.float32) if t.dtype != torch.float32 else t

    @staticmethod
    def device_batch(batch: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        if batch.dtype == torch.uint8:
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

    def device_batch(self, batch: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        kind = {torch.float16: "unit_float16", torch.bfloat16: "unit_bfloat16",
                torch.float32: "unit_float32"}[dtype]
        return self.normalizer.transform(batch, out=kind)
