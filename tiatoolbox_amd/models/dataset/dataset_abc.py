"""Patch datasets (API of reference ``tiatoolbox/models/dataset/dataset_abc.py:451-531``)."""

from __future__ import annotations

import logging
from pathlib import Path
from typing import Callable

import numpy as np

from tiatoolbox_amd.utils.exceptions import DimensionMismatchError

logger = logging.getLogger("tiatoolbox_amd")

_IMG_SUFFIXES = (".npy", ".jpg", ".jpeg", ".tif", ".tiff", ".png")


def load_img(path: str | Path) -> np.ndarray:
    """Load a patch from ``.npy`` or a PIL-readable image (ref. :117-133)."""
    path = Path(path)
    if path.suffix not in _IMG_SUFFIXES:
        msg = f"Cannot load image data from `{path.suffix}` files."
        raise TypeError(msg)
    if path.suffix == ".npy":
        return np.load(path)
    from PIL import Image

    return np.asarray(Image.open(path).convert("RGB"))


def _is_tensor(obj) -> bool:
    import torch

    return isinstance(obj, torch.Tensor)


class PatchDataset:
    """In-memory (NHWC array / list of arrays) or on-disk (list of paths) patches.

    Validation mirrors the reference (``_check_input_integrity`` :73-107, ``__getitem__``
    :501-531): homogeneous input type, equal image shapes, numeric dtype, and a per-item
    shape check against ``patch_input_shape`` raising :class:`DimensionMismatchError`.
    """

    def __init__(self, inputs, labels: list | None = None, patch_input_shape=None) -> None:
        self.inputs = inputs
        self.labels = labels
        self.patch_input_shape = patch_input_shape
        self._preproc: Callable = self.preproc
        self.data_is_npy_alike = False
        self._check_input_integrity()

    @staticmethod
    def preproc(image: np.ndarray) -> np.ndarray:
        return image

    @property
    def preproc_func(self) -> Callable:
        return self._preproc

    @preproc_func.setter
    def preproc_func(self, func: Callable | None) -> None:
        if func is None:
            self._preproc = self.preproc
        elif callable(func):
            self._preproc = func
        else:
            msg = f"{func} is not callable!"
            raise ValueError(msg)

    def _check_input_integrity(self) -> None:
        msg = "Input must be either a list/array of images or a list of valid image paths."
        if _is_tensor(self.inputs):  # MI355X overload: NHWC batch resident in HBM
            if self.inputs.dim() != 4:  # noqa: PLR2004
                msg = "Each sample must be an array of the form HWC."
                raise ValueError(msg)
            self.data_is_npy_alike = True
            return
        if all(isinstance(v, (Path, str)) for v in self.inputs):
            if any(not Path(v).exists() for v in self.inputs):
                raise ValueError(msg)
            shapes = [load_img(v).shape for v in self.inputs]
        elif all(isinstance(v, np.ndarray) for v in self.inputs):
            shapes = [v.shape for v in self.inputs]
            self.data_is_npy_alike = True
        else:
            raise ValueError(msg)
        if any(len(s) != 3 for s in shapes):
            msg = "Each sample must be an array of the form HWC."
            raise ValueError(msg)
        if len({tuple(s) for s in shapes}) > 1:
            msg = "Images must have the same dimensions."
            raise ValueError(msg)
        if isinstance(self.inputs, np.ndarray) and not np.issubdtype(self.inputs.dtype, np.number):
            msg = "Provided input array is non-numerical."
            raise ValueError(msg)

    def check_shape(self, shape: tuple) -> None:
        if self.patch_input_shape is not None and tuple(shape[:-1]) != tuple(self.patch_input_shape):
            msg = (f"Patch size is not compatible with the model. Expected dimensions "
                   f"{tuple(self.patch_input_shape)}, but got {tuple(shape[:-1])}.")
            logger.error(msg)
            raise DimensionMismatchError(expected_dims=tuple(self.patch_input_shape), actual_dims=tuple(shape[:-1]))

    def raw(self, idx: int) -> np.ndarray:
        patch = self.inputs[idx]
        if _is_tensor(patch):
            patch = patch.cpu().numpy()
        if not self.data_is_npy_alike:
            patch = load_img(patch)
        self.check_shape(patch.shape)
        return patch

    def __len__(self) -> int:
        return len(self.inputs)

    def __getitem__(self, idx: int) -> dict:
        data = {"image": self._preproc(self.raw(idx))}
        if self.labels is not None:
            data["label"] = self.labels[idx]
        return data
