"""Model contract of the engines (API of reference ``tiatoolbox/models/models_abc.py:87-256``)."""

from __future__ import annotations

from abc import ABC, abstractmethod
from pathlib import Path
from typing import Callable

import numpy as np
import torch
from torch import nn


def load_torch_model(model: nn.Module, weights: str | Path) -> nn.Module:
    """Load a state dict saved in single-device mode onto the CPU (ref. :28-44)."""
    state = torch.load(weights, map_location="cpu")
    model.load_state_dict(state, strict=True)
    return model


class ModelABC(ABC, nn.Module):
    """Abstract model: ``forward``, static ``infer_batch``, ``preproc_func`` / ``postproc_func`` hooks."""

    def __init__(self) -> None:
        super().__init__()
        self._postproc = self.postproc
        self._preproc = self.preproc
        self.class_dict = None

    @abstractmethod
    def forward(self, *args, **kwargs):
        ...  # pragma: no cover

    @staticmethod
    @abstractmethod
    def infer_batch(model: nn.Module, batch_data, *, device: str):
        ...  # pragma: no cover

    @staticmethod
    def preproc(image: np.ndarray) -> np.ndarray:
        return image

    @staticmethod
    def postproc(image: np.ndarray) -> np.ndarray:
        return image

    @property
    def preproc_func(self) -> Callable:
        return self._preproc

    @preproc_func.setter
    def preproc_func(self, func: Callable | None) -> None:
        if func is not None and not callable(func):
            msg = f"{func} is not callable!"
            raise ValueError(msg)
        self._preproc = self.preproc if func is None else func

    @property
    def postproc_func(self) -> Callable:
        return self._postproc

    @postproc_func.setter
    def postproc_func(self, func: Callable | None) -> None:
        if func is not None and not callable(func):
            msg = f"{func} is not callable!"
            raise ValueError(msg)
        self._postproc = self.postproc if func is None else func

    def to(self, device=None, dtype: torch.dtype | None = None, *, non_blocking: bool = False,
           memory_format=None):
        """Move / cast the model (ref. :204-237).

        The reference wraps the model in single-process ``nn.DataParallel`` when several GPUs
        are visible; here multi-GPU is one process per GPU (``tiatoolbox_amd.distributed``), so
        ``to`` never wraps.  A ``torch.dtype`` as first argument and ``memory_format`` are passed
        through to ``nn.Module.to``.
        """
        if isinstance(device, torch.dtype):
            device, dtype = None, device
        elif device is None and dtype is None and memory_format is None:
            device = "cpu"  # bare ``model.to()``: the reference's default
        kwargs = {"non_blocking": non_blocking}
        if dtype is not None:
            kwargs["dtype"] = dtype
        if memory_format is not None:
            kwargs["memory_format"] = memory_format
        if device is not None:
            kwargs["device"] = torch.device(device)
        return super().to(**kwargs)

    def load_weights_from_file(self, weights: str | Path):
        saved_state_dict = torch.load(weights, map_location="cpu")
        return super().load_state_dict(saved_state_dict, strict=True)
