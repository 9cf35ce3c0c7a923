"""CPU oracle: tissue maskers (restates ``tiatoolbox/tools/tissuemask.py:75-306``).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).
"""

from __future__ import annotations

import numpy as np

from . import cvref, skref


def _grey(image: np.ndarray) -> np.ndarray:
    if image.ndim == 3 and image.shape[-1] == 3:
        return cvref.rgb2gray_u8(image)
    return image[..., 0] if image.ndim == 3 else image


def threshold_otsu(pixels: np.ndarray) -> float:
    """``skimage.filters.threshold_otsu``: integer images via exact bin counts, floats via 256 bins."""
    pixels = np.asarray(pixels)
    if np.issubdtype(pixels.dtype, np.integer):
        return skref.threshold_otsu_u8(pixels)
    first = pixels.reshape(-1)[0]
    if np.all(pixels == first):
        return first
    counts, edges = np.histogram(pixels.ravel(), bins=256, range=(pixels.min(), pixels.max()))
    centers = (edges[:-1] + edges[1:]) / 2.0
    counts = counts.astype(np.float64)
    w1 = np.cumsum(counts)
    w2 = np.cumsum(counts[::-1])[::-1]
    m1 = np.cumsum(counts * centers) / w1
    m2 = (np.cumsum((counts * centers)[::-1]) / w2[::-1])[::-1]
    var12 = w1[:-1] * w2[1:] * (m1[:-1] - m2[1:]) ** 2
    return centers[int(np.argmax(var12))]


class OtsuTissueMasker:
    """``tools/tissuemask.py:75-164``."""

    def __init__(self) -> None:
        self.threshold = None
        self.fitted = False

    def fit(self, images, masks=None) -> None:  # noqa: ARG002
        images_shape = np.shape(images)
        if len(images_shape) != 4:
            msg = f"Expected 4 dimensional input shape (N, height, width, 3) but received shape of {images_shape}."
            raise ValueError(msg)
        grey_images = [_grey(np.asarray(x)) for x in images]
        pixels = np.concatenate([np.array(g).flatten() for g in grey_images])
        self.threshold = threshold_otsu(pixels)
        self.fitted = True

    def transform(self, images) -> np.ndarray:
        if not self.fitted:
            msg = "Fit must be called before transform."
            raise SyntaxError(msg)
        return np.array([(_grey(np.asarray(image)) < self.threshold).astype(bool) for image in images])

    def fit_transform(self, images, **kwargs) -> np.ndarray:
        self.fit(images, masks=None, **kwargs)
        return self.transform(images)


class MorphologicalMasker(OtsuTissueMasker):
    """``tools/tissuemask.py:167-306``."""

    def __init__(self, *, mpp=None, power=None, kernel_size=None, min_region_size=None) -> None:
        super().__init__()
        self.min_region_size = min_region_size
        if sum(arg is not None for arg in [mpp, power, kernel_size]) > 1:
            msg = "Only one of mpp, power, kernel_size can be given."
            raise ValueError(msg)
        if all(arg is None for arg in [mpp, power, kernel_size]):
            kernel_size = np.array([1, 1])
        if power is not None:
            mpp = 10.0 / float(power)  # objective_power2mpp is approximately 10/power
        if mpp is not None:
            mpp_array = np.array(mpp)
            if mpp_array.size != 2:
                mpp_array = mpp_array.repeat(2)
            kernel_size = np.max([32 / mpp_array, np.array([1, 1])], axis=0)
        kernel_size_array = np.array(kernel_size)
        if kernel_size_array.size != 2:
            kernel_size_array = kernel_size_array.repeat(2)
        self.kernel_size = tuple(np.round(kernel_size_array).astype(int))
        self.kernel = cvref.get_structuring_element_ellipse(self.kernel_size)
        if self.min_region_size is None:
            self.min_region_size = int(np.sum(self.kernel))

    def transform(self, images) -> np.ndarray:
        if not self.fitted:
            msg = "Fit must be called before transform."
            raise SyntaxError(msg)
        results = []
        for image in images:
            image = np.asarray(image)
            gray = cvref.rgb2gray_u8(image) if (image.ndim == 3 and image.shape[-1] == 3) else image
            mask = (gray < self.threshold).astype(np.uint8)
            _, output, stats, _ = cvref.connected_components_with_stats(mask, connectivity=8)
            sizes = stats[1:, -1]
            if mask.ndim == 3:
                mask = mask[..., 0]
            for i, size in enumerate(sizes):
                if size < self.min_region_size:
                    mask[output == i + 1] = 0
            mask = cvref.morphology_ex(mask, "DILATE", self.kernel)
            results.append(mask.astype(bool))
        return np.array(results)
