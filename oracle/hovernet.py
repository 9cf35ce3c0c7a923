"""CPU oracle: HoVer-Net post-processing (restates ``tiatoolbox/models/architecture/hovernet.py``).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  ``proc_np_hv`` follows ``_proc_np_hv``
(:502-616) line by line with the cv2 / skimage primitives restated in ``cvref`` / ``skref`` and
scipy's own ``ndimage.label`` / ``binary_fill_holes``; ``get_instance_info`` follows :618-748
(contour polygons through ``cvref.first_contour``, the Suzuki-Abe restatement of ``cv2.findContours``).
"""

from __future__ import annotations

import math

import numpy as np
from scipy import ndimage

from . import cvref, skref


def proc_np_hv(np_map: np.ndarray, hv_map: np.ndarray, scale_factor: float = 1, *, debug: dict | None = None) -> np.ndarray:
    blb_raw = np_map[..., 0]
    h_dir_raw = hv_map[..., 0]
    v_dir_raw = hv_map[..., 1]

    blb = np.array(blb_raw >= 0.5, dtype=np.int32)
    blb = ndimage.label(blb)[0]
    blb = skref.remove_small_objects_labels(blb, max_size=9)
    blb[blb > 0] = 1

    h_dir = cvref.normalize_minmax_to_f32(h_dir_raw)
    v_dir = cvref.normalize_minmax_to_f32(v_dir_raw)

    ksize = int((20 * scale_factor) + 1)
    obj_size = math.ceil(10 * (scale_factor**2))

    sobel_h = cvref.sobel_f64(h_dir, 1, 0, ksize)
    sobel_v = cvref.sobel_f64(v_dir, 0, 1, ksize)
    if debug is not None:
        debug.update(sobel_h_raw=sobel_h, sobel_v_raw=sobel_v)
    sobel_h = 1 - cvref.normalize_minmax_to_f32(sobel_h)
    sobel_v = 1 - cvref.normalize_minmax_to_f32(sobel_v)

    overall = np.maximum(sobel_h, sobel_v)
    overall = overall - (1 - blb)
    overall[overall < 0] = 0

    dist = (1.0 - overall) * blb
    dist = -cvref.gaussian_blur3_f64(dist)

    overall = np.array(overall >= 0.4, dtype=np.int32)

    marker = blb - overall
    marker[marker < 0] = 0
    marker = ndimage.binary_fill_holes(marker).astype("uint8")
    kernel = cvref.get_structuring_element_ellipse((5, 5))
    marker = cvref.morphology_ex(marker, "OPEN", kernel)
    marker = ndimage.label(marker)[0]
    marker = skref.remove_small_objects_labels(marker, max_size=obj_size - 1)
    if debug is not None:
        debug.update(blb=blb, dist=dist, marker=marker, sobel_h=sobel_h, sobel_v=sobel_v)
    return skref.watershed(dist, markers=marker, mask=blb)


def get_bounding_box(img: np.ndarray) -> np.ndarray:
    rows = np.any(img, axis=1)
    cols = np.any(img, axis=0)
    r_min, r_max = np.where(rows)[0][[0, -1]]
    c_min, c_max = np.where(cols)[0][[0, -1]]
    return np.array([c_min, r_min, c_max + 1, r_max + 1])


def get_instance_info(pred_inst: np.ndarray, pred_type: np.ndarray | None = None,
                      offset: tuple[int, int] = (0, 0)) -> dict:
    """``hovernet.py:618-748``: box, centroid (raw moments of the cropped binary mask + top-left),
    contour (``cvref.first_contour``; instances with fewer than 3 vertices are dropped, :695-699),
    majority type and its probability."""
    offset = np.asarray(offset)
    info = {}
    for inst_id in np.unique(pred_inst)[1:]:
        inst_map = pred_inst == inst_id
        box = get_bounding_box(inst_map)
        tl = box[:2] + offset
        crop = inst_map[box[1]:box[3], box[0]:box[2]].astype(np.uint8)
        ys, xs = np.nonzero(crop)
        m00, m10, m01 = float(crop.sum()), float(xs.sum()), float(ys.sum())
        contour = cvref.first_contour(crop).astype(np.int32)
        if contour.shape[0] < 3:  # noqa: PLR2004
            continue
        contour = contour + tl[None].astype(np.int32)
        centroid = np.array([m10 / m00, m01 / m00]) + tl
        out_box = box.copy()
        out_box[:2] += offset
        out_box[2:] += offset
        info[int(inst_id)] = {"box": out_box, "centroid": centroid, "contours": contour, "prob": None, "type": None}
        if pred_type is not None:
            inst_type = pred_type[box[1]:box[3], box[0]:box[2]][crop.astype(bool)]
            type_list, type_pixels = np.unique(inst_type, return_counts=True)
            pairs = sorted(zip(type_list, type_pixels), key=lambda x: x[1], reverse=True)
            top = pairs[0][0]
            if top == 0 and len(pairs) > 1:
                top = pairs[1][0]
            counts = {v[0]: v[1] for v in pairs}
            info[int(inst_id)]["type"] = int(top)
            info[int(inst_id)]["prob"] = float(counts[top] / (np.sum(crop) + 1.0e-6))
    return info


def synth_maps(n: int, h: int, w: int, seed: int = 0, n_blobs: int = 30, num_types: int = 6):
    """Synthetic HoVer-Net head outputs.  The generator is an INPUT source, not a checker: it lives with the
    other synthetic inputs in ``tiatoolbox_amd.utils.synth`` (benches and profiling scripts take it from there, never from
    ``oracle/``); this name stays for the tests."""
    from tiatoolbox_amd.utils.synth import hover_head_maps

    return hover_head_maps(n, h, w, seed=seed, n_blobs=n_blobs, num_types=num_types)
