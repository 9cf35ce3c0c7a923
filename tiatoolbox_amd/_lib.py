"""ctypes binding of the C-ABI library (``include/tiatoolbox_amd.h``).

The product path has **no CPU fallback**: if the HIP library cannot be loaded, or a call
is made without a GPU tensor, it raises.  ``torch`` is used only for device memory and
streams.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

from . import build as _build

TIA_STATS_STRIDE = 64
ST_CYCLES = 48
ST_STAIN, ST_MAXC, ST_NTISSUE, ST_PLOW, ST_PHIGH = 0, 6, 8, 9, 10
ST_MINPHI, ST_MAXPHI, ST_COV, ST_EVEC, ST_FLAGS, ST_PINV, ST_M, ST_SCALE = 11, 12, 13, 19, 25, 26, 32, 41
FLAG_EMPTY_MASK, FLAG_DEGENERATE = 1, 2
MODE_MACENKO, MODE_FIXED, MODE_VAHADANE, MODE_GIVEN = 0, 1, 2, 3
OUT_U8, OUT_F32, OUT_F64, OUT_UNIT_F16, OUT_UNIT_BF16, OUT_UNIT_F32 = 0, 1, 2, 3, 4, 5
MATH_F64, MATH_F32, MATH_F64_REF = 0, 1, 2

TIA_EINVAL, TIA_ELAUNCH, TIA_ESIZE = -1, -2, -3
_ERRORS = {-1: "TIA_EINVAL (bad argument)", -2: "TIA_ELAUNCH (HIP launch failed)",
           -3: "TIA_ESIZE (size not supported)"}


class StainTables(C.Structure):
    _fields_ = [("od_lut", C.c_double * 256), ("od_lut_f32", C.c_float * 256),
                ("ty", (C.c_int32 * 256) * 3)]


class StainParams(C.Structure):
    _fields_ = [
        ("q_img_lo", C.c_double), ("q_img_hi", C.c_double), ("q_phi_lo", C.c_double),
        ("q_phi_hi", C.c_double), ("q_conc", C.c_double), ("stain_fixed", C.c_double * 6),
        ("target_stain", C.c_double * 6), ("target_maxc", C.c_double * 2), ("y_thr", C.c_int32),
        ("mode", C.c_int32), ("has_target", C.c_int32), ("zero_to_one", C.c_int32),
        ("dl_alpha", C.c_double), ("dl_tol", C.c_double), ("dl_max_iter", C.c_int32), ("dl_seed", C.c_int32),
        ("select_mode", C.c_int32), ("dl_one_kernel", C.c_int32),
    ]


class LabTables(C.Structure):
    _fields_ = [("gamma", C.c_uint16 * 256), ("cbrt", C.c_uint16 * 3072), ("lab_y", C.c_uint16 * 256),
                ("lab_ify", C.c_uint16 * 256), ("inv_gamma", C.c_uint8 * 4096), ("c_fwd", C.c_int32 * 9),
                ("c_inv", C.c_int32 * 9)]


class HipLibraryError(RuntimeError):
    """The HIP extension is missing or a kernel launch failed."""


_LIB = None

_I64, _I32, _P = C.c_int64, C.c_int32, C.c_void_p

_SIGNATURES = {
    "tia_abi_version": ([], C.c_int),
    "tia_stain_stats_workspace_bytes": ([_I64, _I64, _I64], C.c_size_t),
    "tia_stain_stats_workspace_bytes_mode": ([_I64, _I64, _I64, _I32], C.c_size_t),
    "tia_stain_stats_u8": ([_P, _I64, _I64, _I64, _P, C.POINTER(StainParams), _P, _P, C.c_size_t, _P], C.c_int),
    "tia_stain_stats_path": ([_I64, _I64, C.POINTER(StainParams)], C.c_int),
    "tia_stain_apply_u8": ([_P, _I64, _I64, _I64, _P, _P, C.POINTER(C.c_double), _P, _I32, _I32, _P], C.c_int),
    "tia_stain_concentrations_f64": ([_P, _I64, _I64, _I64, _P, _P, _P, _P], C.c_int),
    "tia_stain_augment_u8": ([_P, _I64, _I64, _I64, _P, _P, _P, _I32, _I32, _I32, _P, _I32, _P], C.c_int),
    "tia_luminosity_mask_u8": ([_P, _I64, _I64, _I64, _P, _P, _I32, _I32, _P, _P], C.c_int),
    "tia_rgb2od_u8": ([_P, _I64, _P, _I32, _P, _P], C.c_int),
    "tia_clear_last_error": ([], C.c_int),
    "tia_rgb2gray_u8": ([_P, _I64, _P, _P], C.c_int),
    "tia_hist256_u8": ([_P, _I64, _P, _P], C.c_int),
    "tia_threshold_lt_u8": ([_P, _I64, _I32, _I32, _P, _P], C.c_int),
    "tia_gray_hist_u8": ([_P, _I64, _I32, _P, _P], C.c_int),
    "tia_otsu_threshold_u32": ([_P, _P, _P], C.c_int),
    "tia_otsu_fit_u8": ([_P, _I64, _I32, _P, _P, _P], C.c_int),
    "tia_threshold_lt_dev_u8": ([_P, _I64, _I32, _P, _P, _P], C.c_int),
    "tia_morph_mask_u8": ([_P, _I64, _I64, _I64, _I32, _I32, _P, _I32, _P, _I32, _I32, _P, _P], C.c_int),
    "tia_lut_apply_u8": ([_P, _I64, _I64, _P, _P, _P], C.c_int),
    "tia_box_downsample_u8": ([_P, _I64, _I64, _I64, _I64, _P, _P], C.c_int),
    "tia_ccl_label_i32": ([_P, _I64, _I64, _I64, _I32, _P, _P, _P, _P], C.c_int),
    "tia_label_area_filter_i32": ([_P, _I64, _I64, _I64, _I32, _P, _P], C.c_int),
    "tia_binary_morph_u8": ([_P, _I64, _I64, _I64, _P, _I32, _I32, _P, _P], C.c_int),
    "tia_fill_holes_u8": ([_P, _I64, _I64, _I64, _P, _P, _P], C.c_int),
    "tia_hover_workspace_bytes": ([_I64, _I64, _I64], C.c_size_t),
    "tia_hover_proc_np_hv_f32": ([_P, _P, _I64, _I64, _I64, _I32, _I32, _P, _P, _P, C.c_size_t, _P], C.c_int),
    "tia_hover_proc_np_hv_stages_f32": ([_P, _P, _I64, _I64, _I64, _I32, _I32, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P],
                                        C.c_int),
    "tia_watershed_workspace_bytes": ([_I64, _I64, _I64], C.c_size_t),
    "tia_watershed_blobs_f64": ([_P, _P, _P, _I64, _I64, _I64, _P, _P, C.c_size_t, _P], C.c_int),
    "tia_canvas_row_merge_f32": ([_P, _P, _I64, _I64, _I64, _I64, _I64, _P, _P, _P, _P], C.c_int),
    "tia_canvas_finalize_f32": ([_P, _P, _I64, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _P, _P, _P], C.c_int),
    "tia_gather_patches_u8": ([_P, _I64, _I64, _I64, _P, _I64, _I64, _I64, _I32, _P, _P], C.c_int),
    "tia_lab_hist_u8": ([_P, _I64, _I64, _I64, _P, _P, _P], C.c_int),
    "tia_reinhard_apply_u8": ([_P, _I64, _I64, _I64, _P, _P, _P, _P], C.c_int),
    "tia_reinhard_luts": ([_P, _I64, _P, _P, _P, _P, _P, _P, _P], C.c_int),
    "tia_lab_convert_u8": ([_P, _I64, _P, _I32, _P, _P], C.c_int),
    "tia_reinhard_workspace_bytes": ([_I64, _I64, _I64], C.c_size_t),
    "tia_reinhard_transform_u8": ([_P, _I64, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P], C.c_int),
    "tia_lab_moments_u8": ([_P, _I64, _I64, _I64, _P, _P, _P, _P, _P], C.c_int),
    "tia_conv_pack_weights_f32": ([_P, _I64, _I64, _I64, _I64, _P, _P], C.c_int),
    "tia_conv2d_nhwc_f32": ([_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I32, _P], C.c_int),
    "tia_conv2d_nhwc_f32_ex": ([_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I32, _P],
                               C.c_int),
    "tia_conv3x3_geometry": ([_I64, _I64, _I64, _I64, _I64, _I64, C.POINTER(C.c_int32)], C.c_int),
    "tia_conv2d_thin_nhwc_f32": ([_P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I32, _P], C.c_int),
    "tia_conv1x1_head_nhwc_f32": ([_P, _I64, _P, _P, _P, _P, _I32, _P, _P], C.c_int),
    "tia_conv2d_post_nhwc_f32": ([_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I32,
                                  _P, _P, _P, _P], C.c_int),
    "tia_conv2d_route_f32": ([_I64] * 12, C.c_int),
    "tia_conv_pack_weights_wino_f32": ([_P, _I64, _I64, _P, _P], C.c_int),
    "tia_conv3x3_wino_nhwc_f32": ([_P, _P, _P, _P, _P] + [_I64] * 9 + [_I32, _P], C.c_int),
    "tia_conv1x1_pre_nhwc_f32": ([_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I32, _P], C.c_int),
    "tia_stem_pack_weights_h": ([_P, _I32, _P, _P], C.c_int),
    "tia_stem_conv7x7_pool_nhwc_h": ([_P, _I32, _P, _P, _P, _I32, _I64, _I64, _I64, _P], C.c_int),
    "tia_stem_conv7x7_pool_nhwc": ([_P, _I32, _P, _P, _P, _I32, _P, _I64, _I64, _I64, _P], C.c_int),
    "tia_conv2d_nhwc_h": ([_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I32, _I32, _P], C.c_int),
    "tia_conv_pack_weights_h": ([_P, _I64, _I64, _I64, _I64, _I32, _P, _P], C.c_int),
    "tia_stem_pack_weights_f32": ([_P, _P, _P], C.c_int),
    "tia_scale_shift_act_nhwc_f32": ([_P, _P, _P, _P, _I64, _I64, _I32, _P], C.c_int),
    "tia_scale_shift_act_view_nhwc_f32": ([_P, _I64, _I64, _I64, _P, _P, _P, _I64, _I64, _I64, _I64, _I32, _P], C.c_int),
    "tia_grouped_conv_valid_nhwc_f32": ([_P, _P, _P, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _I64, _P], C.c_int),
    "tia_upsample2x_add_nhwc_f32": ([_P, _P, _I64, _I64, _P, _I64, _I64, _I64, _I64, _P], C.c_int),
    "tia_upsample2x_add_act_nhwc_f32": ([_P, _P, _I64, _I64, _P, _P, _P, _I64, _I64, _I64, _I64, _P], C.c_int),
    "tia_bias_act_nhwc": ([_P, _P, _P, _I64, _I64, _I32, _I32, _P], C.c_int),
    "tia_bias_relu_maxpool_nhwc": ([_P, _P, _I64, _I64, _I64, _I64, _I32, _P, _P], C.c_int),
    "tia_hover_instance_stats": ([_P, _P, _I64, _I64, _I64, _I32, _I32, _P, _P, _P], C.c_int),
    "tia_hover_contour_scan": ([_P, _I64, _I64, _I64, _I32, _P, _P, _P, _P, _P], C.c_int),
    "tia_label_first_pixel_i32": ([_P, _I64, _I64, _I64, _I32, _P, _P, _P], C.c_int),
    "tia_border_trace_u8": ([_P, _I64, _I64, _I64, _P, _I64, _I32, _P, _P, _I64, _P, _P], C.c_int),
    "tia_hover_contour_write": ([_P, _I64, _I64, _I64, _I32, _P, _P, _I64, _P, _P], C.c_int),
}


def lib_path() -> Path:
    """The product library, or -- developer switch -- the variant named by ``TIA_LIB_PATH`` (built with
    ``build.build(defines=..., out=...)``; it must export the same C ABI)."""
    override = os.environ.get("TIA_LIB_PATH")
    return Path(override) if override else _build.LIB_PATH


def load() -> C.CDLL:
    """Load ``libtiatoolbox_amd.so`` (built in-tree by ``__graft_entry__.build()``)."""
    global _LIB  # noqa: PLW0603
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not path.exists():
        msg = (f"{path} not found: the HIP extension is not built. Run "
               "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
               "There is no CPU fallback.")
        raise HipLibraryError(msg)
    try:
        lib = C.CDLL(str(path))
    except OSError as exc:  # missing ROCm runtime etc.
        msg = f"cannot load {path}: {exc}"
        raise HipLibraryError(msg) from exc
    for name, (argtypes, restype) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    _LIB = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = f"{what} failed: {_ERRORS.get(rc, rc)}"
        raise HipLibraryError(msg)


def current_stream() -> C.c_void_p:
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(t, name: str = "tensor") -> None:
    if not t.is_cuda:
        msg = f"{name} must live on the GPU (cuda:N == HIP device); there is no CPU fallback."
        raise HipLibraryError(msg)
