"""CPU oracle: stain extraction / normalisation / augmentation (NumPy restatement).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Follows, function by function:

* ``tiatoolbox/utils/transforms.py:209-256``  (``rgb2od`` / ``od2rgb``)
* ``tiatoolbox/utils/misc.py:261-290,405-444`` (``get_luminosity_tissue_mask`` /
  ``contrast_enhancer``)
* ``tiatoolbox/tools/stainextract.py:13-322``  (helpers + the four extractors)
* ``tiatoolbox/tools/stainnorm.py:19-425``     (``StainNormalizer`` family, factory)
* ``tiatoolbox/tools/stainaugment.py:141-235`` (``StainAugmentor.fit/augment``)
"""

from __future__ import annotations

import numpy as np

from . import cvref, skref


# ----------------------------------------------------------------------------- transforms
def rgb2od(img: np.ndarray) -> np.ndarray:
    """``utils/transforms.py:209-231``.  NOTE: mutates ``img`` in place (zeros -> ones)."""
    mask = img == 0
    img[mask] = 1
    return np.maximum(-1 * np.log(img / 255), 1e-6)


def od2rgb(od: np.ndarray) -> np.ndarray:
    """``utils/transforms.py:234-256``."""
    od = np.maximum(od, 1e-6)
    return (255 * np.exp(-1 * od)).astype(np.uint8)


# ----------------------------------------------------------------------------------- misc
def contrast_enhancer(img: np.ndarray, low_p: int = 2, high_p: int = 98) -> np.ndarray:
    """``utils/misc.py:405-444`` (pinned by ``tests/test_utils.py:882-911``)."""
    if img.dtype != np.uint8:
        msg = "Image should be uint8."
        raise AssertionError(msg)
    img_out = img.copy()
    percentiles = np.array(np.percentile(img_out, (low_p, high_p)))
    p_low, p_high = percentiles[0], percentiles[1]
    if p_low >= p_high:
        p_low, p_high = np.min(img_out), np.max(img_out)
    if p_high > p_low:
        img_out = skref.rescale_intensity(img_out, in_range=(p_low, p_high), out_range=(0.0, 255.0))
    return img_out.astype(np.uint8)


def get_luminosity_tissue_mask(img: np.ndarray, threshold: float) -> np.ndarray:
    """``utils/misc.py:261-290``."""
    img = img.astype("uint8")
    img = contrast_enhancer(img, low_p=2, high_p=98)
    img_lab = cvref.rgb2lab_u8(img)
    l_lab = img_lab[:, :, 0] / 255.0
    tissue_mask = l_lab < threshold
    if tissue_mask.sum() == 0:
        msg = "Empty tissue mask computed."
        raise ValueError(msg)
    return tissue_mask


# --------------------------------------------------------------------------- stainextract
def vectors_in_correct_direction(e_vectors: np.ndarray) -> np.ndarray:
    """``tools/stainextract.py:13-30``."""
    if e_vectors[0, 0] < 0:
        e_vectors[:, 0] *= -1
    if e_vectors[0, 1] < 0:
        e_vectors[:, 1] *= -1
    return e_vectors


def h_and_e_in_right_order(v1: np.ndarray, v2: np.ndarray) -> np.ndarray:
    """``tools/stainextract.py:33-50``."""
    if v1[0] > v2[0]:
        return np.array([v1, v2])
    return np.array([v2, v1])


def dl_output_for_h_and_e(dictionary: np.ndarray) -> np.ndarray:
    """``tools/stainextract.py:53-68``."""
    if dictionary[0, 0] < dictionary[1, 0]:
        return dictionary[[1, 0], :]
    return dictionary


class CustomExtractor:
    """``tools/stainextract.py:71-101``."""

    def __init__(self, stain_matrix: np.ndarray) -> None:
        self.stain_matrix = stain_matrix
        if self.stain_matrix.shape not in [(2, 3), (3, 3)]:
            msg = "Stain matrix must have shape (2, 3) or (3, 3)."
            raise ValueError(msg)

    def get_stain_matrix(self, _: np.ndarray) -> np.ndarray:
        return self.stain_matrix


class RuifrokExtractor:
    """``tools/stainextract.py:104-137``."""

    def __init__(self) -> None:
        self.__stain_matrix = np.array([[0.65, 0.70, 0.29], [0.07, 0.99, 0.11]])

    def get_stain_matrix(self, _: np.ndarray) -> np.ndarray:
        return self.__stain_matrix.copy()


class MacenkoExtractor:
    """``tools/stainextract.py:140-227``."""

    def __init__(self, luminosity_threshold: float = 0.8, angular_percentile: float = 99) -> None:
        self.luminosity_threshold = luminosity_threshold
        self.angular_percentile = angular_percentile

    def get_stain_matrix(self, img: np.ndarray, *, debug: dict | None = None) -> np.ndarray:
        img = img.astype("uint8")
        tissue_mask = get_luminosity_tissue_mask(img, threshold=self.luminosity_threshold).reshape((-1,))
        img_od = rgb2od(img).reshape((-1, 3))
        img_od = img_od[tissue_mask]
        cov = np.cov(img_od, rowvar=False)
        _, eigen_vectors = np.linalg.eigh(cov)
        eigen_vectors = eigen_vectors[:, [2, 1]]
        eigen_vectors = vectors_in_correct_direction(e_vectors=eigen_vectors)
        proj = np.dot(img_od, eigen_vectors)
        phi = np.arctan2(proj[:, 1], proj[:, 0])
        min_phi = np.percentile(phi, 100 - self.angular_percentile)
        max_phi = np.percentile(phi, self.angular_percentile)
        v1 = np.dot(eigen_vectors, np.array([np.cos(min_phi), np.sin(min_phi)]))
        v2 = np.dot(eigen_vectors, np.array([np.cos(max_phi), np.sin(max_phi)]))
        he = h_and_e_in_right_order(v1, v2)
        if debug is not None:
            debug.update(mask=tissue_mask, cov=cov, eigen_vectors=eigen_vectors, min_phi=min_phi,
                         max_phi=max_phi, n_tissue=int(tissue_mask.sum()))
        return he / np.linalg.norm(he, axis=1)[:, None]


class VahadaneExtractor:
    """``tools/stainextract.py:230-322``.  ``random_state`` is an oracle-only addition:
    the reference leaves it unset (run-to-run non-deterministic)."""

    def __init__(self, luminosity_threshold: float = 0.8, regularizer: float = 0.1,
                 random_state: int | None = 0) -> None:
        self.luminosity_threshold = luminosity_threshold
        self.regularizer = regularizer
        self.random_state = random_state

    def get_stain_matrix(self, img: np.ndarray) -> np.ndarray:
        from sklearn.decomposition import DictionaryLearning

        img = img.astype("uint8")
        tissue_mask = get_luminosity_tissue_mask(img, threshold=self.luminosity_threshold).reshape((-1,))
        img_od = rgb2od(img).reshape((-1, 3))
        img_od = img_od[tissue_mask]
        dl = DictionaryLearning(
            n_components=2, alpha=self.regularizer, transform_alpha=self.regularizer,
            fit_algorithm="lars", transform_algorithm="lasso_lars", positive_dict=True,
            verbose=False, max_iter=3, transform_max_iter=1000, random_state=self.random_state,
        )
        dictionary = dl.fit_transform(X=img_od.T).T
        dictionary = dl_output_for_h_and_e(dictionary)
        return dictionary / np.linalg.norm(dictionary, axis=1)[:, None]


# ------------------------------------------------------------------------------ stainnorm
class StainNormalizer:
    """``tools/stainnorm.py:19-113``."""

    def __init__(self) -> None:
        self.extractor = None
        self.stain_matrix_target = None
        self.target_concentrations = None
        self.maxC_target = None
        self.stain_matrix_target_RGB = None

    @staticmethod
    def get_concentrations(img: np.ndarray, stain_matrix: np.ndarray) -> np.ndarray:
        od = rgb2od(img).reshape((-1, 3))
        x, _, _, _ = np.linalg.lstsq(stain_matrix.T, od.T, rcond=-1)
        return x.T

    def fit(self, target: np.ndarray) -> None:
        self.stain_matrix_target = self.extractor.get_stain_matrix(target)
        self.target_concentrations = self.get_concentrations(target, self.stain_matrix_target)
        self.maxC_target = np.percentile(self.target_concentrations, 99, axis=0).reshape((1, 2))
        self.stain_matrix_target_RGB = od2rgb(self.stain_matrix_target)

    def transform_float(self, img: np.ndarray) -> np.ndarray:
        """``transform`` up to (excluding) the final ``astype(uint8)`` truncation."""
        stain_matrix_source = self.extractor.get_stain_matrix(img)
        source_concentrations = self.get_concentrations(img, stain_matrix_source)
        max_c_source = np.percentile(source_concentrations, 99, axis=0).reshape((1, 2))
        source_concentrations *= self.maxC_target / max_c_source
        trans = 255 * np.exp(-1 * np.dot(source_concentrations, self.stain_matrix_target))
        trans[trans > 255] = 255
        trans[trans < 0] = 0
        return trans.reshape(img.shape)

    def transform(self, img: np.ndarray) -> np.ndarray:
        return self.transform_float(img).astype(np.uint8)


class CustomNormalizer(StainNormalizer):
    """``tools/stainnorm.py:116-141``."""

    def __init__(self, stain_matrix: np.ndarray) -> None:
        super().__init__()
        self.extractor = CustomExtractor(stain_matrix)


class RuifrokNormalizer(StainNormalizer):
    """``tools/stainnorm.py:144-166``."""

    def __init__(self) -> None:
        super().__init__()
        self.extractor = RuifrokExtractor()


class MacenkoNormalizer(StainNormalizer):
    """``tools/stainnorm.py:169-192``."""

    def __init__(self) -> None:
        super().__init__()
        self.extractor = MacenkoExtractor()


class VahadaneNormalizer(StainNormalizer):
    """``tools/stainnorm.py:195-219``."""

    def __init__(self, random_state: int | None = 0) -> None:
        super().__init__()
        self.extractor = VahadaneExtractor(random_state=random_state)


class ReinhardNormalizer(StainNormalizer):
    """``tools/stainnorm.py:222-367`` (needs ``cvref.lab2rgb_u8``)."""

    def __init__(self) -> None:
        super().__init__()
        self.target_means = None
        self.target_stds = None

    @staticmethod
    def lab_split(img: np.ndarray) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
        img = img.astype("uint8")
        lab = cvref.rgb2lab_u8(img)
        img_float = lab.astype(np.float32)
        chan1, chan2, chan3 = (img_float[..., 0].copy(), img_float[..., 1].copy(), img_float[..., 2].copy())
        chan1 /= np.asarray(2.55)
        chan2 -= np.asarray(128.0)
        chan3 -= np.asarray(128.0)
        return chan1, chan2, chan3

    @staticmethod
    def merge_back(chan1: np.ndarray, chan2: np.ndarray, chan3: np.ndarray) -> np.ndarray:
        chan1 *= 2.55
        chan2 += 128.0
        chan3 += 128.0
        img = np.clip(np.stack((chan1, chan2, chan3), axis=-1), 0, 255).astype(np.uint8)
        return cvref.lab2rgb_u8(img)

    def get_mean_std(self, img: np.ndarray):
        img = img.astype("uint8")
        chan1, chan2, chan3 = self.lab_split(img)
        m1, sd1 = cvref.mean_std_dev(chan1)
        m2, sd2 = cvref.mean_std_dev(chan2)
        m3, sd3 = cvref.mean_std_dev(chan3)
        return (m1, m2, m3), (sd1, sd2, sd3)

    def fit(self, target: np.ndarray) -> None:
        means, stds = self.get_mean_std(target)
        self.target_means = means
        self.target_stds = stds

    def transform_lab_u8(self, img: np.ndarray) -> np.ndarray:
        """Normalised image as 8-bit Lab (the input of the final LAB2RGB)."""
        chan1, chan2, chan3 = self.lab_split(img)
        means, stds = self.get_mean_std(img)
        norm1 = ((chan1 - means[0]) * (self.target_stds[0] / stds[0])) + self.target_means[0]
        norm2 = ((chan2 - means[1]) * (self.target_stds[1] / stds[1])) + self.target_means[1]
        norm3 = ((chan3 - means[2]) * (self.target_stds[2] / stds[2])) + self.target_means[2]
        norm1 *= 2.55
        norm2 += 128.0
        norm3 += 128.0
        return np.clip(np.stack((norm1, norm2, norm3), axis=-1), 0, 255).astype(np.uint8)

    def transform(self, img: np.ndarray) -> np.ndarray:
        return cvref.lab2rgb_u8(self.transform_lab_u8(img))


def get_normalizer(method_name: str, stain_matrix: np.ndarray | None = None) -> StainNormalizer:
    """``tools/stainnorm.py:370-425`` (ndarray stain matrices only)."""
    name = method_name.lower()
    if name not in ["reinhard", "ruifrok", "macenko", "vahadane", "custom"]:
        msg = "Method not supported."
        raise NotImplementedError(msg)
    if stain_matrix is not None and name != "custom":
        msg = '`stain_matrix` is only defined when using `method_name`="custom".'
        raise ValueError(msg)
    if name == "reinhard":
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
    return CustomNormalizer(np.asarray(stain_matrix))


# --------------------------------------------------------------------------- stainaugment
def stain_augment(img: np.ndarray, stain_matrix: np.ndarray, alpha: np.ndarray, beta: np.ndarray,
                  *, threshold: float = 0.85, augment_background: bool = False) -> np.ndarray:
    """``tools/stainaugment.py:141-206`` with the random ``alpha``/``beta`` injected.

    ``fit``: concentrations of ``img`` w.r.t. ``stain_matrix`` and the luminosity tissue
    mask (``threshold`` 0.85 via ``apply``, :227); ``augment``: per stain ``i``
    ``C[mask, i] = C[mask, i] * alpha[i] + beta[i]`` (all pixels when
    ``augment_background``), then ``uint8(clip(255*exp(-C.S), 0, 255))``.
    """
    img = img.copy()
    source_concentrations = StainNormalizer.get_concentrations(img, stain_matrix)
    n_stains = source_concentrations.shape[1]
    tissue_mask = get_luminosity_tissue_mask(img, threshold=threshold).ravel()
    aug = source_concentrations.copy()
    for i in range(n_stains):
        if augment_background:
            aug[:, i] *= alpha[i]
            aug[:, i] += beta[i]
        else:
            aug[tissue_mask, i] *= alpha[i]
            aug[tissue_mask, i] += beta[i]
    img_augmented = 255 * np.exp(-1 * np.dot(aug, stain_matrix))
    img_augmented = img_augmented.reshape(img.shape)
    img_augmented = np.clip(img_augmented, 0, 255)
    return np.uint8(img_augmented)
