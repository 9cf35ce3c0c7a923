"""Stain-matrix extraction (API of reference ``tiatoolbox/tools/stainextract.py``).

``MacenkoExtractor`` runs entirely on the GPU (``tia_stain_stats_u8``: tissue mask, OD
covariance, eigen-decomposition, exact angular percentiles), and so does ``VahadaneExtractor``
(``TIA_MODE_VAHADANE``: the reference's scikit-learn ``DictionaryLearning`` configuration restated
for the 3-sample problem, one workgroup per patch).
"""

from __future__ import annotations

import logging

import numpy as np

from tiatoolbox_amd import _lib
from tiatoolbox_amd.tools import _stain_device as dev
from tiatoolbox_amd.utils import _tensors

logger = logging.getLogger("tiatoolbox_amd")


def vectors_in_correct_direction(e_vectors: np.ndarray) -> np.ndarray:
    """Flip eigenvectors so their first component is non-negative (ref. :13-30)."""
    for col in (0, 1):
        if e_vectors[0, col] < 0:
            e_vectors[:, col] *= -1
    return e_vectors


def h_and_e_in_right_order(v1: np.ndarray, v2: np.ndarray) -> np.ndarray:
    """Haematoxylin (larger red OD) first (ref. :33-50)."""
    return np.array([v1, v2]) if v1[0] > v2[0] else np.array([v2, v1])


def dl_output_for_h_and_e(dictionary: np.ndarray) -> np.ndarray:
    """Order dictionary-learning rows as H, E (ref. :53-68)."""
    if dictionary[0, 0] < dictionary[1, 0]:
        return dictionary[[1, 0], :]
    return dictionary


class CustomExtractor:
    """User-defined stain matrix (ref. :71-101)."""

    def __init__(self, stain_matrix: np.ndarray) -> None:
        self.stain_matrix = stain_matrix
        if self.stain_matrix.shape not in [(2, 3), (3, 3)]:
            msg = "Stain matrix must have shape (2, 3) or (3, 3)."
            raise ValueError(msg)

    def get_stain_matrix(self, _: np.ndarray) -> np.ndarray:
        return self.stain_matrix

    # batched device protocol used by StainNormalizer -------------------------------------------
    def stats_params(self, **kw) -> _lib.StainParams:
        return dev.make_params(mode=_lib.MODE_FIXED, stain_fixed=self.stain_matrix, **kw)


class RuifrokExtractor:
    """Ruifrok & Johnston constant H&E matrix (ref. :104-137)."""

    def __init__(self) -> None:
        self.__stain_matrix = np.array([[0.65, 0.70, 0.29], [0.07, 0.99, 0.11]])

    def get_stain_matrix(self, _: np.ndarray) -> np.ndarray:
        return self.__stain_matrix.copy()

    def stats_params(self, **kw) -> _lib.StainParams:
        return dev.make_params(mode=_lib.MODE_FIXED, stain_fixed=self.__stain_matrix, **kw)


class MacenkoExtractor:
    """Macenko stain extractor (ref. :140-227), computed per patch on the GPU."""

    def __init__(self, luminosity_threshold: float = 0.8, angular_percentile: float = 99) -> None:
        self.__luminosity_threshold = luminosity_threshold
        self.__angular_percentile = angular_percentile

    def stats_params(self, **kw) -> _lib.StainParams:
        return dev.make_params(mode=_lib.MODE_MACENKO, luminosity_threshold=self.__luminosity_threshold,
                               angular_percentile=self.__angular_percentile, **kw)

    def get_stain_matrix(self, img):
        """(2,3) stain matrix of an HWC image; (N,2,3) for an NHWC batch."""
        batch, kind = _tensors.to_device_batch(img)
        stats = dev.stain_stats(batch, self.stats_params())
        dev.raise_on_flags(stats)
        sm = stats[:, _lib.ST_STAIN:_lib.ST_STAIN + 6].reshape(-1, 2, 3)
        return _tensors.from_device(sm, kind)


class VahadaneExtractor:
    """Vahadane stain extractor (ref. :230-322), computed per patch on the GPU.

    ``tia_stain_stats_u8`` in ``TIA_MODE_VAHADANE`` restates scikit-learn's ``DictionaryLearning`` as the reference
    configures it (:305-316: 2 atoms, ``alpha = transform_alpha = regularizer``, LARS coding, ``positive_dict``,
    ``max_iter=3``) for the 3 x N problem at hand -- X = tissue OD transposed, so the *code* (3 x 2) is the stain
    matrix -- one workgroup per patch.  ``random_state`` only seeds the re-draw of a never-used atom
    (``_update_dict``), which the reference leaves to an unseeded generator.
    """

    def __init__(self, luminosity_threshold: float = 0.8, regularizer: float = 0.1) -> None:
        logger.warning(
            "Vahadane stain extraction/normalization algorithms are unstable "
            "after the update to `dictionary learning` algorithm in "
            "scikit-learn > v0.23.0 (see issue #382). Please be advised and "
            "consider using other stain extraction (normalization) algorithms.",
            stacklevel=2,
        )
        self.__luminosity_threshold = luminosity_threshold
        self.__regularizer = regularizer
        self.random_state = None  # reference leaves DictionaryLearning unseeded (:305-315)
        self.max_iter = 3         # :313

    def stats_params(self, **kw) -> _lib.StainParams:
        seed = 0 if self.random_state is None else int(self.random_state) & 0x7fffffff
        return dev.make_params(mode=_lib.MODE_VAHADANE, luminosity_threshold=self.__luminosity_threshold,
                               dl_alpha=self.__regularizer, dl_max_iter=self.max_iter, dl_seed=seed, **kw)

    def get_stain_matrix(self, img):
        """(2,3) stain matrix of an HWC image; (N,2,3) for an NHWC batch."""
        batch, kind = _tensors.to_device_batch(img)
        stats = dev.stain_stats(batch, self.stats_params())
        dev.raise_on_flags(stats)
        sm = stats[:, _lib.ST_STAIN:_lib.ST_STAIN + 6].reshape(-1, 2, 3)
        return _tensors.from_device(sm, kind)
