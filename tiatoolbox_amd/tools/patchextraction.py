"""Patch grid generation and tissue-mask filtering (API of reference
``tiatoolbox/tools/patchextraction.py:356-613``: the two static methods the engines use).

Pure index arithmetic on a few thousand coordinates (host side); the mask-area test uses a
summed-area table of the low-resolution mask instead of a Python loop over coordinates.
"""

from __future__ import annotations

import numpy as np


def _invalid_shape(shape: np.ndarray) -> bool:
    return (not np.issubdtype(shape.dtype, np.integer)) or np.size(shape) > 2 or bool(np.any(shape < 0))  # noqa: PLR2004


class PatchExtractor:
    """Namespace for the coordinate helpers (the iterator classes of the reference are out of scope)."""

    @staticmethod
    def get_coordinates(patch_output_shape=None, image_shape=None, patch_input_shape=None, stride_shape=None, *,
                        input_within_bound: bool = False, output_within_bound: bool = False):
        """Input (and output) patch bounds ``[x0, y0, x1, y1]`` tiling an image (ref. :487-613).

        Output patches tile from 0 with ``stride`` (``ceil(dim/stride)`` per axis, x fastest); input
        patches are centred on them (``(in - out)//2`` offset, may be negative).
        """
        return_output_bound = patch_output_shape is not None
        image = np.array(image_shape)
        p_in = np.array(patch_input_shape)
        if patch_output_shape is None:
            output_within_bound = False
            p_out = p_in
        else:
            p_out = np.array(patch_output_shape)
        stride = np.array(stride_shape)
        for name, arr in (("image_shape", image), ("patch_input_shape", p_in), ("patch_output_shape", p_out),
                          ("stride_shape", stride)):
            if _invalid_shape(arr):
                msg = f"Invalid `{name}` value {arr}."
                raise ValueError(msg)
        if np.any(p_in < p_out):
            msg = f"`patch_input_shape` must larger than `patch_output_shape` {p_in} must > {p_out}."
            raise ValueError(msg)
        if np.any(stride < 1):
            msg = f"`stride_shape` value {stride} must > 1."
            raise ValueError(msg)
        xs = np.arange(0, int(np.ceil(image[0] / stride[0]) * stride[0]), stride[0])
        ys = np.arange(0, int(np.ceil(image[1] / stride[1]) * stride[1]), stride[1])
        xv, yv = np.meshgrid(xs, ys)
        out_tl = np.stack([xv.flatten(), yv.flatten()], axis=-1)
        out_br = out_tl + p_out[None]
        in_tl = out_tl - ((p_in - p_out) // 2)[None]
        in_br = in_tl + p_in[None]
        drop = np.zeros(in_tl.shape[0], dtype=bool)
        if output_within_bound:
            drop |= np.any(out_br > image[None], axis=1)
        if input_within_bound:
            drop |= np.any(in_br > image[None], axis=1)
            drop |= np.any(in_tl < 0, axis=1)
        in_bounds = np.concatenate([in_tl[~drop], in_br[~drop]], axis=-1)
        out_bounds = np.concatenate([out_tl[~drop], out_br[~drop]], axis=-1)
        return (in_bounds, out_bounds) if return_output_bound else in_bounds

    @staticmethod
    def filter_coordinates(mask_reader, coordinates_list: np.ndarray, wsi_shape, min_mask_ratio: float = 0,
                           func=None) -> np.ndarray:
        """Keep coordinates whose footprint on the tissue mask is positive enough (ref. :356-461).

        ``mask_reader`` is anything with an ``img`` attribute holding the 2-D mask (the reference
        insists on a ``VirtualWSIReader``).  Coordinates are scaled into mask space in float32 and
        truncated to int32, exactly as the reference does.
        """
        if not hasattr(mask_reader, "img"):
            msg = "`mask_reader` should be wsireader.VirtualWSIReader."
            raise TypeError(msg)
        if not isinstance(coordinates_list, np.ndarray) or not np.issubdtype(coordinates_list.dtype, np.integer):
            msg = "`coordinates_list` should be ndarray of integer type."
            raise ValueError(msg)
        if coordinates_list.shape[-1] != 4:  # noqa: PLR2004
            msg = "`coordinates_list` must be of shape [N, 4]."
            raise ValueError(msg)
        if not 0 <= min_mask_ratio <= 1:
            msg = "`min_mask_ratio` must be between 0 and 1."
            raise ValueError(msg)
        tissue_mask = np.asarray(mask_reader.img)
        scale = np.array(tissue_mask.shape[1::-1]) / np.array(wsi_shape)
        sc = coordinates_list.copy().astype(np.float32)
        sc[:, [0, 2]] *= scale[0]
        sc[:, [0, 2]] = np.clip(sc[:, [0, 2]], 0, tissue_mask.shape[1])
        sc[:, [1, 3]] *= scale[1]
        sc[:, [1, 3]] = np.clip(sc[:, [1, 3]], 0, tissue_mask.shape[0])
        ic = sc.astype(np.int32)
        if func is not None:
            return np.array([func(tissue_mask, c) for c in ic.tolist()])
        # summed-area table of the (small) mask
        sat = np.zeros((tissue_mask.shape[0] + 1, tissue_mask.shape[1] + 1), dtype=np.int64)
        sat[1:, 1:] = np.cumsum(np.cumsum(tissue_mask != 0, axis=0), axis=1)
        x0, y0, x1, y1 = ic[:, 0], ic[:, 1], ic[:, 2], ic[:, 3]
        x1c, y1c = np.maximum(x1, x0), np.maximum(y1, y0)
        area = (x1c - x0).astype(np.int64) * (y1c - y0)
        pos = sat[y1c, x1c] - sat[y0, x1c] - sat[y1c, x0] + sat[y0, x0]
        return ((pos == area) | (pos > area * min_mask_ratio)) & (pos > 0) & (area > 0)
