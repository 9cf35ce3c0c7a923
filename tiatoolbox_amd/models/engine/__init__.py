"""Engines (module names of reference ``tiatoolbox/models/engine/__init__.py``, for the covered path)."""

from __future__ import annotations

import importlib

__all__ = ["engine_abc", "io_config", "multi_task_segmentor", "nucleus_instance_segmentor", "patch_predictor",
           "semantic_segmentor"]


def __getattr__(name: str):
    if name in __all__:
        return importlib.import_module(f"{__name__}.{name}")
    msg = f"module {__name__!r} has no attribute {name!r}"
    raise AttributeError(msg)
