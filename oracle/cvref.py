"""Restatements of the OpenCV primitives the reference's hot path calls.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  OpenCV (``opencv-python>=4.6.0``,
reference ``requirements/requirements.txt:19``) is not vendored under
``/root/reference`` and is not installable here, so these follow the published
algorithm of OpenCV 4.x ``modules/imgproc/src/color_lab.cpp`` / ``color_yuv``/
``color_rgb`` (8-bit fixed-point paths).  **Parity with the real library is
unpinned**: verify every table when ``cv2`` is importable.

Reference call sites: ``tiatoolbox/utils/misc.py:281`` (RGB2LAB for the luminosity
mask), ``tiatoolbox/tools/stainnorm.py:309,340`` (Reinhard), ``tiatoolbox/tools/
tissuemask.py:129,160,291`` (RGB2GRAY).
"""

from __future__ import annotations

import functools

import numpy as np

GAMMA_SHIFT = 3
LAB_SHIFT = 12
LAB_SHIFT2 = LAB_SHIFT + GAMMA_SHIFT
LAB_CBRT_TAB_SIZE_B = 256 * 3 // 2 * (1 << GAMMA_SHIFT)  # 3072

# sRGB -> XYZ (D65) as used by OpenCV (color_lab.cpp: sRGB2XYZ_D65, D65 white point)
_SRGB2XYZ_D65 = np.array(
    [0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227],
    dtype=np.float64,
)
_D65 = np.array([0.950456, 1.0, 1.088754], dtype=np.float64)


def _cv_round(x: np.ndarray) -> np.ndarray:
    """cvRound: round half to even."""
    return np.rint(x).astype(np.int64)


def _cv_cbrt_f32(x32: np.ndarray) -> np.ndarray:
    """cv::cbrt(softfloat): exponent split + quartic rational polynomial (f64), cast to f32.

    Follows OpenCV ``modules/core/src/mathfuncs_core``/``softfloat.cpp`` ``cubeRoot``.
    """
    x32 = np.asarray(x32, dtype=np.float32)
    bits = x32.view(np.int32).astype(np.int64)
    ix = bits & 0x7FFFFFFF
    s = bits & 0x80000000
    ex = (ix >> 23) - 127
    # C remainder (truncation toward zero), then forced negative
    shx = np.fmod(ex, 3).astype(np.int64)
    shx = shx - np.where(shx >= 0, 3, 0)
    ex3 = (ex - shx) // 3  # exact division
    frbits = (ix & ((1 << 23) - 1)) | ((shx + 127) << 23)
    fr = frbits.astype(np.int32).view(np.float32).astype(np.float64)
    num = ((((45.2548339756803022511987494 * fr + 192.2798368355061050458134625) * fr
             + 119.1654824285581628956914143) * fr + 13.43250139086239872172837314) * fr
           + 0.1636161226585754240958355063)
    den = ((((14.80884093219134573786480845 * fr + 151.9714051044435648658557668) * fr
             + 168.5254414101568283957668343) * fr + 33.9905941350215598754191872) * fr + 1.0)
    r32 = (num / den).astype(np.float32)
    rbits = r32.view(np.int32).astype(np.int64)
    out = (rbits + (ex3 << 23) + s)
    out = np.where(ix != 0, out, 0)
    return (out & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


@functools.lru_cache(maxsize=1)
def lab_tables() -> dict[str, np.ndarray]:
    """Build OpenCV's 8-bit Lab tables (``initLabTabs`` + ``RGB2Lab_b`` ctor)."""
    f32 = np.float32
    i = np.arange(256)
    x = (i.astype(f32) / f32(255.0)).astype(f32)
    xd = x.astype(np.float64)
    g = np.where(xd <= 0.04045, xd / 12.92, np.power((xd + 0.055) / 1.055, 2.4))
    g32 = g.astype(f32)
    srgb_gamma = _cv_round((f32(255 * (1 << GAMMA_SHIFT)) * g32).astype(f32))
    linear_gamma = i * (1 << GAMMA_SHIFT)

    j = np.arange(LAB_CBRT_TAB_SIZE_B)
    cb_scale = (f32(1.0) / (f32(255.0) * f32(1 << GAMMA_SHIFT))).astype(f32)
    xj = (cb_scale * j.astype(f32)).astype(f32)
    lthresh = (f32(216.0) / f32(24389.0)).astype(f32)
    lscale = (f32(841.0) / f32(108.0)).astype(f32)
    lbias = (f32(16.0) / f32(116.0)).astype(f32)
    # mulAdd(x, lscale, lbias) is a fused multiply-add in f32
    lin = (xj.astype(np.float64) * np.float64(lscale) + np.float64(lbias)).astype(f32)
    with np.errstate(all="ignore"):
        cb = _cv_cbrt_f32(xj)
    val = np.where(xj < lthresh, lin, cb).astype(f32)
    cbrt_tab = _cv_round((f32(1 << LAB_SHIFT2) * val).astype(f32))

    coeffs = np.empty(9, dtype=np.int64)
    for r in range(3):
        for c in range(3):
            coeffs[r * 3 + c] = int(np.rint((1 << LAB_SHIFT) * _SRGB2XYZ_D65[r * 3 + c] / _D65[r]))
    return {
        "srgb_gamma": srgb_gamma.astype(np.int64),
        "linear_gamma": linear_gamma.astype(np.int64),
        "cbrt": cbrt_tab.astype(np.int64),
        "coeffs": coeffs,
    }


L_SCALE = (116 * 255 + 50) // 100  # 296
L_SHIFT = -((16 * 255 * (1 << LAB_SHIFT2) + 50) // 100)


def _descale(x: np.ndarray, n: int) -> np.ndarray:
    return (x + (1 << (n - 1))) >> n


def rgb2lab_u8(img: np.ndarray) -> np.ndarray:
    """``cv2.cvtColor(img, cv2.COLOR_RGB2LAB)`` for uint8 input (``RGB2Lab_b``)."""
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.shape[-1] != 3:
        msg = "rgb2lab_u8 expects HxWx3 uint8"
        raise ValueError(msg)
    t = lab_tables()
    tab, cb, c = t["srgb_gamma"], t["cbrt"], t["coeffs"]
    r = tab[img[..., 0]]
    g = tab[img[..., 1]]
    b = tab[img[..., 2]]
    fx = cb[_descale(r * c[0] + g * c[1] + b * c[2], LAB_SHIFT)]
    fy = cb[_descale(r * c[3] + g * c[4] + b * c[5], LAB_SHIFT)]
    fz = cb[_descale(r * c[6] + g * c[7] + b * c[8], LAB_SHIFT)]
    big_l = _descale(L_SCALE * fy + L_SHIFT, LAB_SHIFT2)
    a = _descale(500 * (fx - fy) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2)
    bb = _descale(200 * (fy - fz) + 128 * (1 << LAB_SHIFT2), LAB_SHIFT2)
    out = np.stack([big_l, a, bb], axis=-1)
    return np.clip(out, 0, 255).astype(np.uint8)


def lab_l_from_y_table() -> np.ndarray:
    """L (uint8) as a function of the descaled Y index 0..3071 (helper for host code)."""
    t = lab_tables()
    fy = t["cbrt"]
    return np.clip(_descale(L_SCALE * fy + L_SHIFT, LAB_SHIFT2), 0, 255).astype(np.uint8)


def rgb2gray_u8(img: np.ndarray) -> np.ndarray:
    """``cv2.cvtColor(img, cv2.COLOR_RGB2GRAY)`` uint8: 15-bit fixed point (OpenCV 4.x).

    ``(R*9798 + G*19235 + B*3735 + 2^14) >> 15`` (``color_rgb``: ``RGB2Gray<uchar>``,
    ``gray_shift = 15``, coefficients ``RY15=9798, GY15=19235, BY15=3735``).
    """
    img = np.asarray(img)
    r = img[..., 0].astype(np.int64)
    g = img[..., 1].astype(np.int64)
    b = img[..., 2].astype(np.int64)
    return ((r * 9798 + g * 19235 + b * 3735 + (1 << 14)) >> 15).astype(np.uint8)


def mean_std_dev(chan: np.ndarray) -> tuple[float, float]:
    """``cv2.meanStdDev`` on one channel: f64 mean and *population* standard deviation."""
    x = np.asarray(chan, dtype=np.float64).ravel()
    n = x.size
    s = float(x.sum())
    sq = float((x * x).sum())
    mean = s / n
    var = max(sq / n - mean * mean, 0.0)
    return mean, float(np.sqrt(var))


# ------------------------------------------------------------------------------ morphology / CC
def get_structuring_element_ellipse(ksize: tuple[int, int]) -> np.ndarray:
    """``cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (w, h))`` (OpenCV ``morph.dispatch.cpp``).

    Row ``i``: ``dy = i - r``; ``dx = round(c*sqrt((r*r - dy*dy)/r^2))``; ones on
    ``[max(c-dx,0), min(c+dx+1, w))`` with ``r = h//2``, ``c = w//2``; ``(1,1)`` is a rectangle.
    Call sites: ``tools/tissuemask.py:268``, ``models/architecture/hovernet.py:605``.
    """
    w, h = int(ksize[0]), int(ksize[1])
    elem = np.zeros((h, w), dtype=np.uint8)
    if (w, h) == (1, 1):
        elem[:] = 1
        return elem
    r, c = h // 2, w // 2
    inv_r2 = 1.0 / (r * r) if r else 0.0
    for i in range(h):
        dy = i - r
        if abs(dy) <= r:
            dx = int(np.rint(c * np.sqrt((r * r - dy * dy) * inv_r2)))
            j1, j2 = max(c - dx, 0), min(c + dx + 1, w)
            elem[i, j1:j2] = 1
    return elem


def _morph(mask: np.ndarray, kernel: np.ndarray, *, erode: bool) -> np.ndarray:
    """Binary erode/dilate with anchor at the kernel centre; OpenCV default border
    (erode: outside counts as 1, dilate: outside counts as 0)."""
    kh, kw = kernel.shape
    ay, ax = kh // 2, kw // 2
    h, w = mask.shape
    src = mask != 0
    out = np.ones((h, w), bool) if erode else np.zeros((h, w), bool)
    for i in range(kh):
        for j in range(kw):
            if not kernel[i, j]:
                continue
            dy, dx = i - ay, j - ax
            shifted = np.full((h, w), erode, dtype=bool)
            ys, ye = max(0, -dy), min(h, h - dy)
            xs, xe = max(0, -dx), min(w, w - dx)
            if ys < ye and xs < xe:
                shifted[ys:ye, xs:xe] = src[ys + dy:ye + dy, xs + dx:xe + dx]
            out = (out & shifted) if erode else (out | shifted)
    return out.astype(np.uint8)


def morphology_ex(mask: np.ndarray, op: str, kernel: np.ndarray) -> np.ndarray:
    """``cv2.morphologyEx`` for binary uint8 images: ``"DILATE"``, ``"ERODE"``, ``"OPEN"``."""
    mask = np.asarray(mask)
    if mask.ndim == 3 and mask.shape[-1] == 1:
        mask = mask[..., 0]
    if op == "DILATE":
        return _morph(mask, kernel, erode=False)
    if op == "ERODE":
        return _morph(mask, kernel, erode=True)
    if op == "OPEN":
        return _morph(_morph(mask, kernel, erode=True), kernel, erode=False)
    if op == "CLOSE":
        return _morph(_morph(mask, kernel, erode=False), kernel, erode=True)
    raise NotImplementedError(op)


def connected_components_with_stats(mask: np.ndarray, connectivity: int = 8):
    """``cv2.connectedComponentsWithStats``: ``(n, labels, stats, centroids)``; ``stats[:, -1]`` = area.
    Labels follow the raster order of each component's first pixel (``scipy.ndimage.label``)."""
    from scipy import ndimage

    mask = np.asarray(mask)
    if mask.ndim == 3 and mask.shape[-1] == 1:
        mask = mask[..., 0]
    structure = np.ones((3, 3), int) if connectivity == 8 else None
    labels, n = ndimage.label(mask != 0, structure=structure)
    stats = np.zeros((n + 1, 5), dtype=np.int32)
    stats[:, 4] = np.bincount(labels.ravel(), minlength=n + 1)
    return n + 1, labels.astype(np.int32), stats, None


# names used by tests/golden/_refshim.py to stand in for the cv2 functions
def getStructuringElement(shape, ksize):  # noqa: N802
    assert shape == "ELLIPSE"
    return get_structuring_element_ellipse(tuple(int(k) for k in ksize))


def morphologyEx(src, op, kernel):  # noqa: N802
    return morphology_ex(src, op, kernel)


def connectedComponentsWithStats(mask, connectivity=8):  # noqa: N802
    return connected_components_with_stats(mask, connectivity)


# ------------------------------------------------------------------------ filtering (HoVer-Net)
def normalize_minmax_to_f32(src: np.ndarray) -> np.ndarray:
    """``cv2.normalize(src, None, alpha=0, beta=1, norm_type=NORM_MINMAX, dtype=CV_32F)``.

    ``scale = 1/(max-min)`` (0 when ``max-min < DBL_EPSILON``), ``shift = -min*scale``, then
    ``convertTo``: float32 sources are scaled in float32 arithmetic, float64 sources in float64
    and then rounded to float32.  No fused multiply-add is assumed (unverifiable here; see
    DESIGN.md "parity unpinned").  Call sites: ``models/architecture/hovernet.py:547-590``.
    """
    src = np.asarray(src)
    smin, smax = float(src.min()), float(src.max())
    rng = smax - smin
    scale = 1.0 / rng if rng > np.finfo(np.float64).eps else 0.0
    shift = 0.0 - smin * scale
    if src.dtype == np.float32:
        a, b = np.float32(scale), np.float32(shift)
        return (src * a).astype(np.float32) + b
    return (src.astype(np.float64) * scale + shift).astype(np.float32)


def _reflect101(idx: np.ndarray, n: int) -> np.ndarray:
    """OpenCV BORDER_REFLECT_101 index mapping (gfedcb|abcdefgh|gfedcba)."""
    if n == 1:
        return np.zeros_like(idx)
    period = 2 * (n - 1)
    idx = np.mod(idx, period)
    return np.where(idx >= n, period - idx, idx)


def sobel_kernels(ksize: int, dx: int, dy: int) -> tuple[np.ndarray, np.ndarray]:
    """``cv2.getDerivKernels`` for Sobel with ``ksize > 3`` (``getSobelKernels``): integer kernels
    built by repeated [1,1] smoothing and [-1,1] differencing; returned as float64 (kx, ky)."""
    out = []
    for order in (dx, dy):
        ker = np.zeros(ksize + 1, dtype=np.int64)
        ker[0] = 1
        for _ in range(ksize - order - 1):
            oldval = ker[0]
            for j in range(1, ksize + 1):
                newval = ker[j] + ker[j - 1]
                ker[j - 1] = oldval
                oldval = newval
        for _ in range(order):
            oldval = -ker[0]
            for j in range(1, ksize + 1):
                newval = ker[j - 1] - ker[j]
                ker[j - 1] = oldval
                oldval = newval
        out.append(ker[:ksize].astype(np.float64))
    return out[0], out[1]


def _row_filter(src: np.ndarray, kx: np.ndarray) -> np.ndarray:
    """OpenCV ``RowFilter<ST,double>``: ``s = kx[0]*S[0]; s += kx[k]*S[k]`` for k ascending (f64)."""
    h, w = src.shape
    anchor = len(kx) // 2
    cols = _reflect101(np.arange(w)[None, :] + (np.arange(len(kx))[:, None] - anchor), w)
    s64 = src.astype(np.float64)
    acc = kx[0] * s64[:, cols[0]]
    for k in range(1, len(kx)):
        acc = acc + kx[k] * s64[:, cols[k]]
    return acc


def _column_filter(buf: np.ndarray, ky: np.ndarray, *, symmetric: bool) -> np.ndarray:
    """OpenCV ``SymmColumnFilter``: centre tap then paired taps ``ky[k]*(S[+k] +/- S[-k])``."""
    h, w = buf.shape
    c = len(ky) // 2
    rows = np.arange(h)
    if symmetric:
        acc = ky[c] * buf[rows, :] + 0.0
    else:
        acc = np.zeros_like(buf)
    for k in range(1, c + 1):
        up = buf[_reflect101(rows + k, h), :]
        dn = buf[_reflect101(rows - k, h), :]
        acc = acc + ky[c + k] * ((up + dn) if symmetric else (up - dn))
    return acc


def sobel_f64(src: np.ndarray, dx: int, dy: int, ksize: int) -> np.ndarray:
    """``cv2.Sobel(src, cv2.CV_64F, dx, dy, ksize=ksize)`` (``ksize > 3``; BORDER_REFLECT_101).

    Separable: generic row filter with ``kx`` (sequential taps, f64), then the (anti)symmetric
    column filter with ``ky``.  Call sites: ``hovernet.py:568-569``.
    """
    kx, ky = sobel_kernels(ksize, dx, dy)
    buf = _row_filter(np.asarray(src), kx)
    return _column_filter(buf, ky, symmetric=(dy % 2 == 0))


def gaussian_blur3_f64(src: np.ndarray) -> np.ndarray:
    """``cv2.GaussianBlur(src, (3, 3), 0)`` for float64: fixed kernel [1/4, 1/2, 1/4], separable,
    symmetric small filters ``S[0]*k0 + (S[-1] + S[1])*k1``, BORDER_REFLECT_101 (``hovernet.py:598``)."""
    src = np.asarray(src, dtype=np.float64)
    h, w = src.shape
    k0, k1 = 0.5, 0.25
    xs = np.arange(w)
    row = src * k0 + (src[:, _reflect101(xs - 1, w)] + src[:, _reflect101(xs + 1, w)]) * k1
    ys = np.arange(h)
    return row * k0 + (row[_reflect101(ys - 1, h), :] + row[_reflect101(ys + 1, h), :]) * k1


def normalize(src, dst=None, alpha=0, beta=1, norm_type=None, dtype=None):  # noqa: ARG001
    return normalize_minmax_to_f32(src)


def Sobel(src, ddepth, dx, dy, ksize=3):  # noqa: N802, ARG001
    return sobel_f64(src, dx, dy, ksize)


def GaussianBlur(src, ksize, sigma):  # noqa: N802, ARG001
    assert tuple(ksize) == (3, 3) and sigma == 0
    return gaussian_blur3_f64(src)


# ----------------------------------------------------------------------------- Lab -> RGB (8-bit)
LAB_BASE_SHIFT = 14
LAB_BASE = 1 << LAB_BASE_SHIFT
INV_GAMMA_SHIFT = 12
INV_GAMMA_TAB_SIZE = 1 << INV_GAMMA_SHIFT
MIN_AB_VALUE = -8145
_XYZ2SRGB_D65 = np.array([3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311])


def _trunc_div(a, b: int):
    """C integer division (truncation toward zero) for NumPy integer arrays."""
    a = np.asarray(a, dtype=np.int64)
    q = np.abs(a) // b
    return np.where(a < 0, -q, q)


def ab_to_xz(i):
    """``abToXZ_b[i - minABvalue]`` of OpenCV's ``initLabTabs`` computed directly (C integer arithmetic)."""
    i = np.asarray(i, dtype=np.int64)
    lin = _trunc_div(i * 108, 841) - (LAB_BASE * 16 // 116 * 108 // 841)
    cub = _trunc_div(_trunc_div(i * i, LAB_BASE) * i, LAB_BASE)
    return np.where(i <= 3390, lin, cub)


@functools.lru_cache(maxsize=1)
def lab2rgb_tables() -> dict[str, np.ndarray]:
    """``LabToYF_b``, ``sRGBInvGammaTab_b`` and the 12-bit XYZ->sRGB coefficients (``Lab2RGBinteger``)."""
    f32 = np.float32
    i = np.arange(256)
    base = f32(LAB_BASE)
    y_lo = np.rint((i * LAB_BASE * 20 * 9).astype(f32) / f32(17 * 29 * 29 * 29))
    ify_lo = np.rint(base * (f32(16) / f32(116) + (i * 5).astype(f32) / f32(3 * 17 * 29)).astype(f32))
    fy = ((i * 100 * LAB_BASE).astype(f32) / f32(255 * 116) + f32(16 * LAB_BASE) / f32(116)).astype(f32)
    ify_hi = np.rint(fy)
    y_hi = np.rint(((fy * fy).astype(f32) * fy).astype(f32) / f32(float(LAB_BASE) * LAB_BASE))
    lab_to_y = np.where(i <= 20, y_lo, y_hi).astype(np.int64)
    lab_to_ify = np.where(i <= 20, ify_lo, ify_hi).astype(np.int64)
    x = (np.arange(INV_GAMMA_TAB_SIZE).astype(f32) / f32(INV_GAMMA_TAB_SIZE - 1)).astype(np.float64)
    inv = np.where(x <= 0.0031308, x * 12.92, 1.055 * np.power(x, 1.0 / 2.4) - 0.055)
    inv_gamma = np.rint((f32(255.0) * inv.astype(f32)).astype(f32)).astype(np.int64)
    coeffs = np.array([int(np.rint((1 << LAB_SHIFT) * _XYZ2SRGB_D65[r * 3 + k] * _D65[k])) for r in range(3) for k in range(3)],
                      dtype=np.int64)
    return {"y": lab_to_y, "ify": lab_to_ify, "inv_gamma": inv_gamma, "coeffs": coeffs}


def lab2rgb_u8(lab: np.ndarray) -> np.ndarray:
    """``cv2.cvtColor(lab, cv2.COLOR_LAB2RGB)`` for uint8 input: OpenCV 4.x ``Lab2RGBinteger``
    (14-bit fixed point, inverse-gamma table of 4096 entries).  **Parity unpinned** (see module doc)."""
    lab = np.asarray(lab)
    t = lab2rgb_tables()
    ll = lab[..., 0].astype(np.int64)
    aa = lab[..., 1].astype(np.int64)
    bb = lab[..., 2].astype(np.int64)
    y = t["y"][ll]
    ify = t["ify"][ll]
    adiv = ((5 * aa * 53687 + (1 << 7)) >> 13) - 128 * LAB_BASE // 500
    bdiv = ((bb * 41943 + (1 << 4)) >> 9) - 128 * LAB_BASE // 200 + 1
    x = ab_to_xz(ify + adiv)
    z = ab_to_xz(ify - bdiv)
    c = t["coeffs"]
    shift = LAB_SHIFT + (LAB_BASE_SHIFT - INV_GAMMA_SHIFT)
    out = []
    for r in range(3):
        v = _descale(c[r * 3] * x + c[r * 3 + 1] * y + c[r * 3 + 2] * z, shift)
        out.append(t["inv_gamma"][np.clip(v, 0, INV_GAMMA_TAB_SIZE - 1)])
    return np.clip(np.stack(out, axis=-1), 0, 255).astype(np.uint8)


# ------------------------------------------------------------------ findContours (Suzuki-Abe)
# 8-neighbourhood chain codes of OpenCV (x right, y down): 0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE
_CODE_DXY = ((1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1))


def _follow_border(f: np.ndarray, x0: int, y0: int, nbd: int, *, is_hole: bool, simple: bool = True) -> list:
    """Border following of one border (Suzuki & Abe 1985, step 3; OpenCV ``icvFetchContour``).

    ``f`` is the zero-padded int image (0 background, 1 unvisited foreground, +-k visited), ``x0,y0``
    the padded coordinates of the start pixel.  Marks ``-nbd`` where the right neighbour is a
    zero pixel examined by the search, ``nbd`` on other yet unmarked pixels.  With ``simple`` a point is
    emitted only where the chain code changes (``CHAIN_APPROX_SIMPLE``).  Returned points are in
    padded coordinates.
    """
    s_end = s = 0 if is_hole else 4
    while True:  # first non-zero neighbour clockwise from the start direction
        s = (s - 1) & 7
        x1, y1 = x0 + _CODE_DXY[s][0], y0 + _CODE_DXY[s][1]
        if f[y1, x1] != 0 or s == s_end:
            break
    if f[y1, x1] == 0:  # isolated pixel
        f[y0, x0] = -nbd
        return [(x0, y0)]
    pts = []
    x3, y3 = x0, y0
    prev_s = s ^ 4
    while True:
        s_end = s
        while True:  # counter-clockwise search from the direction after the one we came from
            s += 1
            x4, y4 = x3 + _CODE_DXY[s & 7][0], y3 + _CODE_DXY[s & 7][1]
            if f[y4, x4] != 0:
                break
        passed_east = s > 8  # direction 0 (== 8 unwrapped) was examined and found zero
        s &= 7
        if passed_east:
            f[y3, x3] = -nbd
        elif f[y3, x3] == 1:
            f[y3, x3] = nbd
        if s != prev_s or not simple:
            pts.append((x3, y3))
            prev_s = s
        if (x4, y4) == (x0, y0) and (x3, y3) == (x1, y1):
            break
        x3, y3 = x4, y4
        s = (s + 4) & 7
    return pts


def find_contours_tree(mask: np.ndarray, *, simple: bool = True) -> list[dict]:
    """All borders of a binary image with their topology, in raster order of discovery.

    ``cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE)`` restated from Suzuki & Abe's
    algorithm 1 (the one OpenCV implements).  Each entry: ``points`` (k,2) int32 ``(x, y)``,
    ``is_hole``, ``parent`` (index into the list, -1 = frame).  **Parity unpinned** (no cv2 here);
    structural known answers are asserted in ``tests/test_oracle_golden.py``.
    """
    m = np.asarray(mask) != 0
    h, w = m.shape
    f = np.zeros((h + 2, w + 2), dtype=np.int64)
    f[1:-1, 1:-1] = m
    borders: list[dict] = []  # border k+2 <-> borders[k]; border 1 = frame (a hole border)
    for y in range(1, h + 1):
        lnbd = 1
        for x in range(1, w + 2):
            p, prev = f[y, x], f[y, x - 1]
            start = None
            if p == 1 and prev == 0:
                start, is_hole = (x, y), False
            elif p == 0 and prev >= 1:
                start, is_hole = (x - 1, y), True
                if prev > 1:
                    lnbd = int(prev)
            if start is not None:
                # parent (Suzuki & Abe, table 1): same kind as the last border met -> share its parent,
                # otherwise that border is the parent; the frame (index -1) behaves as a hole border
                lnbd_hole = True if lnbd == 1 else borders[lnbd - 2]["is_hole"]
                lnbd_parent = -1 if lnbd == 1 else borders[lnbd - 2]["parent"]
                parent = lnbd_parent if lnbd_hole == is_hole else lnbd - 2
                nbd = len(borders) + 2
                pts = _follow_border(f, start[0], start[1], nbd, is_hole=is_hole, simple=simple)
                borders.append({"points": np.array(pts, dtype=np.int32).reshape(-1, 2) - 1,
                                "is_hole": is_hole, "parent": parent})
                p = f[y, x]
            if p not in (0, 1):
                lnbd = abs(int(p))
    return borders


def find_contours(mask: np.ndarray, *, simple: bool = True) -> list[np.ndarray]:
    """``cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE | CHAIN_APPROX_NONE)[0]`` as a list of (k, 2) arrays.

    Order = OpenCV's: every new border is linked at the *front* of its parent's child list and the tree is enumerated
    in pre-order (node, its children, next sibling), so siblings come out in reverse order of discovery.  Call site
    needing the whole list: ``hovernetplus.py:222-226`` (``_get_layer_info``).  Parity unpinned (no cv2 here).
    """
    tree = find_contours_tree(mask, simple=simple)
    children: dict[int, list[int]] = {}
    for i, b in enumerate(tree):
        children.setdefault(b["parent"], []).append(i)
    out: list[np.ndarray] = []
    stack = list(children.get(-1, []))  # popping from the end = last discovered first
    while stack:
        i = stack.pop()
        out.append(tree[i]["points"])
        stack.extend(children.get(i, []))  # its children come next, again last discovered first
    return out


def first_contour(mask: np.ndarray) -> np.ndarray:
    """``cv2.findContours(mask, RETR_TREE, CHAIN_APPROX_SIMPLE)[0][0]`` squeezed to (k, 2).

    OpenCV links every new border at the *front* of its parent's child list
    (``cvInsertNodeIntoTree``) and enumerates the tree in pre-order, so element 0 is the top-level
    outer border that was discovered **last** in raster order.  Call site: ``hovernet.py:685-692``.
    """
    tops = [b for b in find_contours_tree(mask) if not b["is_hole"] and b["parent"] == -1]
    return tops[-1]["points"]
