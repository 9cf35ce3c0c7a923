"""Stain-matrix extraction (API of reference ``tiatoolbox/tools/stainextract.py``).

``MacenkoExtractor`` runs entirely on the GPU (``tia_stain_stats_u8``: tissue mask, OD
covariance, eigen-decomposition, exact angular percentiles).  ``VahadaneExtractor`` uses
the GPU for the tissue mask / OD conversion and scikit-learn's ``DictionaryLearning`` for
the (tiny, sequential, 3-sample) dictionary solve, exactly as the reference does.
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
    """Vahadane stain extractor (ref. :230-322)."""

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

    def get_stain_matrix(self, img: np.ndarray) -> np.ndarray:
        from sklearn.decomposition import DictionaryLearning

        from tiatoolbox_amd.utils.misc import get_luminosity_tissue_mask
        from tiatoolbox_amd.utils.transforms import rgb2od

        batch, _ = _tensors.to_device_batch(img)
        mask = get_luminosity_tissue_mask(batch, threshold=self.__luminosity_threshold)[0].reshape(-1)
        img_od = rgb2od(batch)[0].reshape(-1, 3)[mask].cpu().numpy()
        dl = DictionaryLearning(
            n_components=2, alpha=self.__regularizer, transform_alpha=self.__regularizer,
            fit_algorithm="lars", transform_algorithm="lasso_lars", positive_dict=True, verbose=False,
            max_iter=3, transform_max_iter=1000, random_state=self.random_state,
        )
        dictionary = dl.fit_transform(X=img_od.T).T
        dictionary = dl_output_for_h_and_e(dictionary)
        return dictionary / np.linalg.norm(dictionary, axis=1)[:, None]
