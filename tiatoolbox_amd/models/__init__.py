"""Models package (names exported like reference ``tiatoolbox/models/__init__.py``, for the covered path)."""

from __future__ import annotations

import importlib

_EXPORTS = {
    "HoVerNet": "architecture.hovernet",
    "PatchDataset": "dataset.dataset_abc",
    "IOInstanceSegmentorConfig": "engine.io_config",
    "IOPatchPredictorConfig": "engine.io_config",
    "IOSegmentorConfig": "engine.io_config",
    "ModelIOConfigABC": "engine.io_config",
    "MultiTaskSegmentor": "engine.multi_task_segmentor",
    "NucleusInstanceSegmentor": "engine.multi_task_segmentor",
    "PatchPredictor": "engine.patch_predictor",
    "DeepFeatureExtractor": "engine.deep_feature_extractor",
    "SemanticSegmentor": "engine.semantic_segmentor",
}
_SUBMODULES = ("architecture", "dataset", "engine", "models_abc")

__all__ = [*_EXPORTS, *_SUBMODULES]


def __getattr__(name: str):
    if name in _SUBMODULES:
        return importlib.import_module(f"{__name__}.{name}")
    if name in _EXPORTS:
        return getattr(importlib.import_module(f"{__name__}.{_EXPORTS[name]}"), name)
    msg = f"module {__name__!r} has no attribute {name!r}"
    raise AttributeError(msg)
