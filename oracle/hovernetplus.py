"""CPU oracle: HoVerNet+ post-processing (restates ``tiatoolbox/models/architecture/hovernetplus.py``).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  ``proc_ls`` follows ``_proc_ls`` (:140-187) and
``get_layer_info`` follows ``_get_layer_info`` (:189-247) line by line, with ``cv2.morphologyEx`` /
``cv2.findContours`` / ``skimage.morphology.remove_small_objects`` restated in ``cvref`` / ``skref``.  Nuclei use
``oracle.hovernet.proc_np_hv(scale_factor=0.5)`` (:358).
"""

from __future__ import annotations

import numpy as np

from . import cvref, skref
from .hovernet import get_bounding_box


def proc_ls(ls_map: np.ndarray) -> np.ndarray:
    ls_map = np.squeeze(ls_map)
    ls_map = np.around(ls_map).astype("uint8")
    min_size = 20000
    kernel_size = 20
    epith_all = np.where(ls_map >= 2, 1, 0).astype("uint8")
    mask = np.where(ls_map >= 1, 1, 0).astype("uint8")
    epith_all = epith_all > 0
    epith_mask = skref.remove_small_objects(epith_all, max_size=min_size - 1).astype("uint8")
    epith_edited = (epith_mask * ls_map).astype("uint8")
    epith_edited_open = np.zeros_like(epith_edited).astype("uint8")
    kernel = np.ones((kernel_size, kernel_size))
    for i in [3, 2, 4]:
        tmp = np.where(epith_edited == i, 1, 0).astype("uint8")
        ep_open = cvref.morphology_ex(tmp, "CLOSE", kernel)
        ep_open = cvref.morphology_ex(ep_open, "OPEN", kernel)
        epith_edited_open[ep_open == 1] = i
    mask_open = cvref.morphology_ex(mask, "CLOSE", kernel)
    mask_open = cvref.morphology_ex(mask_open, "OPEN", kernel).astype("uint8")
    ls_map = mask_open.copy()
    for i in range(2, 5):
        ls_map[epith_edited_open == i] = i
    return ls_map.astype("uint8")


def get_layer_info(pred_layer: np.ndarray, offset: tuple[int, int] = (0, 0)) -> dict:
    layer_list = np.unique(pred_layer)
    layer_list = np.delete(layer_list, np.where(layer_list == 0))
    info = {}
    count = 1
    offset = np.asarray(offset)
    for type_class in layer_list:
        layer = np.where(pred_layer == type_class, 1, 0).astype("uint8")
        bounding_box = get_bounding_box(layer)
        for contour in cvref.find_contours(layer, simple=False):  # RETR_TREE, CHAIN_APPROX_NONE
            if contour.shape[0] < 3:  # noqa: PLR2004
                continue
            box = bounding_box.copy()
            box[:2] = box[:2] + offset
            box[2:] = box[2:] + offset
            info[count] = {"box": box, "contours": contour + offset, "type": type_class}
            count += 1
    return info


def synth_layer_map(h: int, w: int, seed: int = 0) -> np.ndarray:
    """Synthetic ``ls`` head output (float32, values 0..4): horizontal tissue bands with wavy borders (background 0,
    connective tissue 1, three epithelial layers 2..4), speckle, small islands and holes -- exercises the size filter
    and the 20x20 closings/openings of ``_proc_ls``."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    wave = 9.0 * np.sin(xx / 23.0 + seed) + 5.0 * np.sin(xx / 7.0)
    edges = np.array([0.12, 0.30, 0.52, 0.70, 0.86]) * h
    ls = np.zeros((h, w), np.int64)
    for k, (lo, hi) in enumerate(zip(edges[:-1], edges[1:])):
        ls[(yy >= lo + wave) & (yy < hi + wave)] = [1, 2, 3, 4][k]
    speck = rng.random((h, w)) < 0.04
    ls[speck] = rng.integers(0, 5, int(speck.sum()))
    for _ in range(12):  # islands / holes of assorted sizes
        cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(3, 18)
        ls[(yy - cy) ** 2 + (xx - cx) ** 2 < r * r] = rng.integers(0, 5)
    return (ls + rng.normal(0, 0.1, (h, w))).clip(0, 4).astype(np.float32)[..., None]
