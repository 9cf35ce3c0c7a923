"""Known answers of the third-party primitives the oracle restates (cv2 / scikit-image are absent here, so these pin the
restatements to PUBLISHED behaviour instead of to themselves): values quoted in the libraries' documentation and test
suites, or derivable in closed form from the documented definition.  What stays unverifiable is listed in DESIGN.md 2."""

from __future__ import annotations

import numpy as np
import pytest

from oracle import cvref, skref


def test_rgb2lab_8bit_cube_corners_and_grey_ramp():
    """cv2.cvtColor(RGB2LAB) on uint8: L*255/100, a+128, b+128 of CIE Lab(D65) (OpenCV docs, "RGB <-> CIE L*a*b*").
    Corners: the primaries as printed in countless OpenCV answers, the secondaries from the CIE formulas (+-1: the 8-bit
    path works in 12/15-bit fixed point).  Greys have a = b = 128 exactly and L = round(2.55 L*)."""
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255]]], dtype=np.uint8)
    exp = np.array([[[255, 128, 128], [0, 128, 128], [136, 208, 195], [224, 42, 211], [82, 207, 20]]])
    assert np.array_equal(cvref.rgb2lab_u8(px), exp)
    sec = np.array([[[255, 255, 0], [0, 255, 255], [255, 0, 255]]], dtype=np.uint8)
    # CIE L*a*b* of yellow (97.14, -21.55, 94.48), cyan (91.11, -48.09, -14.13), magenta (60.32, 98.23, -60.82)
    cie = np.array([[97.14, -21.55, 94.48], [91.11, -48.09, -14.13], [60.32, 98.23, -60.82]])
    want = np.stack([cie[:, 0] * 2.55, cie[:, 1] + 128, cie[:, 2] + 128], axis=1)
    assert np.abs(cvref.rgb2lab_u8(sec)[0].astype(float) - want).max() <= 1.0
    grey = np.arange(0, 256, 5, dtype=np.uint8)
    lab = cvref.rgb2lab_u8(np.stack([grey] * 3, axis=-1)[None])[0]
    assert np.all(lab[:, 1] == 128) and np.all(lab[:, 2] == 128)
    lin = np.where(grey / 255 <= 0.04045, grey / 255 / 12.92, ((grey / 255 + 0.055) / 1.055) ** 2.4)
    lstar = np.where(lin > 216 / 24389, 116 * np.cbrt(lin) - 16, 24389 / 27 * lin)
    assert np.abs(lab[:, 0].astype(float) - lstar * 2.55).max() <= 1.0
    back = cvref.lab2rgb_u8(lab[None])[0]                     # greys survive the 8-bit round trip to +-1
    assert np.abs(back.astype(int) - grey[:, None]).max() <= 1


def test_rgb2gray_8bit_weights():
    """cv2 RGB2GRAY: Y = 0.299 R + 0.587 G + 0.114 B, rounded (OpenCV docs): the primaries give 76, 150, 29."""
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [10, 200, 90]]], dtype=np.uint8)
    got = cvref.rgb2gray_u8(px)[0]
    assert got.tolist()[:5] == [76, 150, 29, 255, 0]
    assert abs(int(got[5]) - round(0.299 * 10 + 0.587 * 200 + 0.114 * 90)) <= 1


def test_sobel_on_impulse_and_ramp():
    """cv2.Sobel is the separable binomial-smoothing x first-difference filter of getDerivKernels (unnormalised):
    ksize 3 and 5 kernels as printed in the OpenCV docs; for any odd k the response to a unit ramp along the derivative
    axis is sum(smooth) * sum_i i d_i = 2^(k-1) * 2^(k-2) in the interior (REFLECT_101 only touches the border)."""
    kx, ky = cvref.sobel_kernels(3, 1, 0)
    assert kx.tolist() == [-1, 0, 1] and ky.tolist() == [1, 2, 1]
    kx, ky = cvref.sobel_kernels(5, 1, 0)
    assert kx.tolist() == [-1, -2, 0, 2, 1] and ky.tolist() == [1, 4, 6, 4, 1]
    imp = np.zeros((9, 9))
    imp[4, 4] = 1.0
    resp = cvref.sobel_f64(imp, 1, 0, 3)
    assert resp[3:6, 3:6].tolist() == [[1, 0, -1], [2, 0, -2], [1, 0, -1]]  # correlation with [[-1,0,1],[-2,0,2],[-1,0,1]]
    for k in (11, 21):
        ramp = np.tile(np.arange(64, dtype=np.float64), (64, 1))
        gx = cvref.sobel_f64(ramp, 1, 0, k)
        gy = cvref.sobel_f64(ramp, 0, 1, k)
        c = slice(k, 64 - k)
        assert np.all(gx[c, c] == 2.0 ** (k - 1) * 2.0 ** (k - 2)) and np.all(gy[c, c] == 0.0)


def test_gaussian_blur3_and_normalize():
    """cv2.GaussianBlur((3,3), 0) uses the fixed kernel [0.25, 0.5, 0.25] (getGaussianKernel docs: small fixed kernels for
    ksize <= 7, sigma <= 0); cv2.normalize(NORM_MINMAX, 0, 1) maps min -> 0, max -> 1 linearly."""
    imp = np.zeros((7, 7))
    imp[3, 3] = 16.0
    assert cvref.gaussian_blur3_f64(imp)[2:5, 2:5].tolist() == [[1, 2, 1], [2, 4, 2], [1, 2, 1]]
    x = np.array([[2.0, 4.0], [6.0, 10.0]], np.float32)
    assert cvref.normalize_minmax_to_f32(x).tolist() == [[0.0, 0.25], [0.5, 1.0]]
    assert not cvref.normalize_minmax_to_f32(np.full((3, 3), 7.0, np.float32)).any()  # zero range -> scale 0


def test_structuring_element_and_morphology_anchor():
    """getStructuringElement(MORPH_ELLIPSE, (5,5)) as printed in the OpenCV docs; with an even all-ones kernel the
    default anchor is ksize // 2, so dilating one pixel by ones((4,4)) covers offsets -2..+1 (what makes closing with
    even kernels shift-asymmetric, hovernetplus.py:167-183)."""
    assert cvref.get_structuring_element_ellipse((5, 5)).tolist() == [
        [0, 0, 1, 0, 0], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [1, 1, 1, 1, 1], [0, 0, 1, 0, 0]]
    m = np.zeros((9, 9), np.uint8)
    m[4, 4] = 1
    d = cvref.morphology_ex(m, "DILATE", np.ones((4, 4)))
    ys, xs = np.nonzero(d)
    # dst(y, x) = max over element offsets of src(y + dy - 2, x + dx - 2): the set pixels are 4 - (dy - 2) for dy = 0..3
    assert (ys.min(), ys.max(), xs.min(), xs.max()) == (3, 6, 3, 6)
    # erode uses the SAME offsets (the documented formulas of cv2.erode / cv2.dilate differ only in min/max), so closing
    # with an even element moves an isolated pixel by one: the well-known one-pixel shift of even-sized kernels
    e = cvref.morphology_ex(d, "ERODE", np.ones((4, 4)))
    assert np.argwhere(e).tolist() == [[5, 5]] and np.array_equal(e, cvref.morphology_ex(m, "CLOSE", np.ones((4, 4))))


def test_skimage_documented_examples():
    """``remove_small_objects`` docstring example (scikit-image morphology/misc.py; min_size=6 == max_size=5);
    ``threshold_otsu`` of a two-valued image returns the lower value (documented: pixels > threshold are foreground)."""
    a = np.array([[0, 0, 0, 1, 0], [1, 1, 1, 0, 0], [1, 1, 1, 0, 1]], bool)
    b = skref.remove_small_objects(a, max_size=5)
    assert b.tolist() == [[False, False, False, False, False], [True, True, True, False, False], [True, True, True, False, False]]
    lab = np.array([[1, 1, 0, 2], [1, 0, 0, 2], [0, 3, 0, 2]])
    assert skref.remove_small_objects(lab, max_size=2).tolist() == [[1, 1, 0, 2], [1, 0, 0, 2], [0, 0, 0, 2]]
    two = np.array([[10, 10, 200], [200, 10, 200]], np.uint8)
    assert skref.threshold_otsu_u8(two) == 10
    assert skref.threshold_otsu_u8(np.full((4, 4), 37, np.uint8)) == 37
    ramp = np.arange(256, dtype=np.uint8)[None].repeat(4, 0)      # uniform histogram: the split sits in the middle
    assert skref.threshold_otsu_u8(ramp) in (127, 128)
    assert skref.rescale_intensity(np.array([0, 5, 10]), (0, 10), (0.0, 255.0)).tolist() == [0.0, 127.5, 255.0]


def test_watershed_documented_behaviour():
    """skimage.segmentation.watershed basics (documented): markers keep their labels, a flat image splits a line between
    two markers at equal distance in favour of the earlier-queued marker, the mask is never crossed, unreachable mask
    regions stay 0."""
    img = np.zeros((1, 9))
    mk = np.zeros((1, 9), np.int32)
    mk[0, 0], mk[0, 8] = 1, 2
    out = skref.watershed(img, mk, np.ones((1, 9), bool))
    assert out[0].tolist() == [1, 1, 1, 1, 1, 2, 2, 2, 2] or out[0].tolist() == [1, 1, 1, 1, 2, 2, 2, 2, 2]
    mask = np.ones((1, 9), bool)
    mask[0, 4] = False
    out = skref.watershed(img, mk, mask)
    assert out[0].tolist() == [1, 1, 1, 1, 0, 2, 2, 2, 2]
    hill = np.array([[0.0, 1.0, 5.0, 1.0, 0.0]])
    mk = np.array([[1, 0, 0, 0, 2]], np.int32)
    assert skref.watershed(hill, mk, np.ones((1, 5), bool))[0].tolist() in ([1, 1, 1, 2, 2], [1, 1, 2, 2, 2])


@pytest.mark.gpu
def test_hip_lab_gray_known_answers():
    """The same published constants on the HIP kernels (not via the oracle)."""
    import torch

    from tiatoolbox_amd.tools import _img_device as img
    from tiatoolbox_amd.tools.reinhard import lab_convert

    px = np.array([[[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [128, 128, 128]]]], dtype=np.uint8)
    lab = lab_convert(torch.from_numpy(px).cuda(), 0).cpu().numpy()[0, 0]
    assert lab[:5].tolist() == [[255, 128, 128], [0, 128, 128], [136, 208, 195], [224, 42, 211], [82, 207, 20]]
    assert lab[5, 1] == 128 and lab[5, 2] == 128 and abs(int(lab[5, 0]) - 137) <= 1
    grey = img.rgb2gray(torch.from_numpy(px).cuda()).cpu().numpy()[0, 0]
    assert grey.tolist() == [255, 0, 76, 150, 29, 128]
