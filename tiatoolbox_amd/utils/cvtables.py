"""Host-side constant tables uploaded to the GPU once (tiny; computed with NumPy).

* OD look-up tables for ``rgb2od`` (reference ``utils/transforms.py:229-231``): a uint8
  image has only 256 distinct optical densities, so ``-log(x/255)`` becomes a 256-entry
  table evaluated here with NumPy's own ``log`` (bit-identical to what the reference
  computes per pixel).
* OpenCV 8-bit ``COLOR_RGB2LAB`` fixed-point tables (``RGB2Lab_b``; gamma table, cube-root
  table, 12-bit XYZ coefficients).  OpenCV is not vendored by the reference; constants are
  those of OpenCV 4.x ``modules/imgproc/src/color_lab.cpp``.
"""

from __future__ import annotations

import functools

import numpy as np

GAMMA_SHIFT, LAB_SHIFT = 3, 12
LAB_SHIFT2 = LAB_SHIFT + GAMMA_SHIFT
CBRT_TAB_SIZE = 256 * 3 // 2 * (1 << GAMMA_SHIFT)
L_SCALE = (116 * 255 + 50) // 100
L_SHIFT = -((16 * 255 * (1 << LAB_SHIFT2) + 50) // 100)

_XYZ = (0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227)
_WHITE = (0.950456, 1.0, 1.088754)


def od_lut() -> np.ndarray:
    """``max(-log(max(v,1)/255), 1e-6)`` for v = 0..255 (float64)."""
    v = np.arange(256, dtype=np.uint8)
    v[0] = 1
    return np.maximum(-1 * np.log(v / 255), 1e-6)


def _cbrt32(x32: np.ndarray) -> np.ndarray:
    """OpenCV's float cube root (exponent/3 + quartic rational polynomial evaluated in f64)."""
    bits = x32.astype(np.float32).view(np.int32).astype(np.int64)
    ix = bits & 0x7FFFFFFF
    ex = (ix >> 23) - 127
    shx = np.fmod(ex, 3).astype(np.int64)
    shx -= np.where(shx >= 0, 3, 0)
    ex3 = (ex - shx) // 3
    fr = ((ix & 0x7FFFFF) | ((shx + 127) << 23)).astype(np.int32).view(np.float32).astype(np.float64)
    num = ((((45.2548339756803022511987494 * fr + 192.2798368355061050458134625) * fr
             + 119.1654824285581628956914143) * fr + 13.43250139086239872172837314) * fr
           + 0.1636161226585754240958355063)
    den = ((((14.80884093219134573786480845 * fr + 151.9714051044435648658557668) * fr
             + 168.5254414101568283957668343) * fr + 33.9905941350215598754191872) * fr + 1.0)
    rb = (num / den).astype(np.float32).view(np.int32).astype(np.int64) + (ex3 << 23)
    rb = np.where(ix != 0, rb, 0)
    return (rb & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


@functools.lru_cache(maxsize=1)
def lab_tables() -> dict[str, np.ndarray]:
    f32 = np.float32
    x = (np.arange(256).astype(f32) / f32(255)).astype(np.float64)
    gam = np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4).astype(f32)
    gamma_tab = np.rint((f32(255 * (1 << GAMMA_SHIFT)) * gam).astype(f32)).astype(np.int64)
    scale = f32(1.0) / (f32(255.0) * f32(1 << GAMMA_SHIFT))
    xj = (scale * np.arange(CBRT_TAB_SIZE).astype(f32)).astype(f32)
    thresh, lsc, lbias = f32(216.0) / f32(24389.0), f32(841.0) / f32(108.0), f32(16.0) / f32(116.0)
    lin = (xj.astype(np.float64) * np.float64(lsc) + np.float64(lbias)).astype(f32)
    with np.errstate(all="ignore"):
        val = np.where(xj < thresh, lin, _cbrt32(xj)).astype(f32)
    cbrt_tab = np.rint((f32(1 << LAB_SHIFT2) * val).astype(f32)).astype(np.int64)
    coeffs = np.array([round((1 << LAB_SHIFT) * _XYZ[i] / _WHITE[i // 3]) for i in range(9)], dtype=np.int64)
    return {"gamma": gamma_tab, "cbrt": cbrt_tab, "coeffs": coeffs}


def _descale(x: np.ndarray, n: int) -> np.ndarray:
    return (x + (1 << (n - 1))) >> n


def l_of_y() -> np.ndarray:
    """8-bit Lab L as a function of the descaled Y table index (0..3071)."""
    fy = lab_tables()["cbrt"]
    return np.clip(_descale(L_SCALE * fy + L_SHIFT, LAB_SHIFT2), 0, 255)


def y_threshold(luminosity_threshold: float) -> int:
    """Tissue ``<=>`` descaled Y index ``< y_thr`` (``L/255.0 < threshold``, misc.py:282-283)."""
    lum = l_of_y().astype(np.float64) / 255.0
    below = lum < luminosity_threshold
    n = int(np.count_nonzero(below))
    if not np.array_equal(below, np.arange(below.size) < n):  # L(Y) is monotone
        msg = "Lab L table is not monotone"
        raise AssertionError(msg)
    return n


def ty_tables() -> np.ndarray:
    """``ty[c][v] = C[3+c] * gamma_tab[v]``: the Y row of RGB2Lab_b, one table per channel."""
    t = lab_tables()
    return np.stack([t["coeffs"][3 + c] * t["gamma"] for c in range(3)]).astype(np.int32)


# ---------------------------------------------------------------- OpenCV Lab2RGBinteger tables
_XYZ2RGB = (3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311)
LAB_BASE = 1 << 14
INV_GAMMA_TAB_SIZE = 1 << 12


@functools.lru_cache(maxsize=1)
def lab_inverse_tables() -> dict[str, np.ndarray]:
    """``LabToYF_b`` (y, ify), ``sRGBInvGammaTab_b`` and the 12-bit XYZ->sRGB coefficients."""
    f32 = np.float32
    i = np.arange(256)
    y_lo = np.rint((i * LAB_BASE * 20 * 9).astype(f32) / f32(17 * 29 * 29 * 29))
    ify_lo = np.rint(f32(LAB_BASE) * (f32(16) / f32(116) + (i * 5).astype(f32) / f32(3 * 17 * 29)).astype(f32))
    fy = ((i * 100 * LAB_BASE).astype(f32) / f32(255 * 116) + f32(16 * LAB_BASE) / f32(116)).astype(f32)
    y_hi = np.rint(((fy * fy).astype(f32) * fy).astype(f32) / f32(float(LAB_BASE) * LAB_BASE))
    x = (np.arange(INV_GAMMA_TAB_SIZE).astype(f32) / f32(INV_GAMMA_TAB_SIZE - 1)).astype(np.float64)
    inv = np.where(x <= 0.0031308, x * 12.92, 1.055 * np.power(x, 1.0 / 2.4) - 0.055)
    return {
        "y": np.where(i <= 20, y_lo, y_hi).astype(np.int64),
        "ify": np.where(i <= 20, ify_lo, np.rint(fy)).astype(np.int64),
        "inv_gamma": np.rint((f32(255.0) * inv.astype(f32)).astype(f32)).astype(np.int64),
        "coeffs": np.array([round((1 << LAB_SHIFT) * _XYZ2RGB[r * 3 + k] * _WHITE[k]) for r in range(3) for k in range(3)],
                           dtype=np.int64),
    }
