"""Dataset helpers (names of reference ``tiatoolbox/models/dataset/__init__.py``, for the covered path)."""

from tiatoolbox_amd.models.dataset.classification import predefined_preproc_func
from tiatoolbox_amd.models.dataset.dataset_abc import PatchDataset

__all__ = ["PatchDataset", "predefined_preproc_func"]
