"""Pins of the oracle's third-party restatements against the REAL libraries, wherever those exist.

``oracle/cvref.py`` (OpenCV) and ``oracle/skref.py`` (scikit-image) restate primitives of libraries that cannot be
installed in the build container; the reference's goldens were produced with the reference's Python logic running on these
restatements (``tests/golden/_refshim.py``).  This module closes the loop on any machine that does have the real wheels:
every test ``importorskip``s its library and compares the restatement with the real call on the same seeded inputs the golden
fixtures and the GPU tests use -- bit-exact for integer / label / contour work, exact equality for the float64 filters
(same tap order), documented tolerance otherwise.  ``scripts/probe_env.py`` records which of the libraries a box has
(``profiles/r04_env_probe.txt`` for the GPU box of round 4).

The reference pins the same layer through real PNGs and real cv2 (``/root/reference/tests/test_stainnorm.py:71-165``,
mean-abs < 1e-2); here the comparison is primitive by primitive and exact.

scipy IS installed in the build container (a real dependency of the reference: ``hovernet.py:15``), so the morphology and
labelling restatements are additionally pinned against ``scipy.ndimage`` on every run.
"""

from __future__ import annotations

import numpy as np
import pytest

from oracle import cvref, skref
from tiatoolbox_amd.utils import synth


def _blobs(seed: int, shape=(96, 128), p: float = 0.55) -> np.ndarray:
    rng = np.random.default_rng(seed)
    m = rng.random(shape) < p
    # smooth into blobs with holes: majority of the 3x3 neighbourhood, twice
    for _ in range(2):
        pad = np.pad(m, 1)
        s = sum(pad[i:i + shape[0], j:j + shape[1]] for i in range(3) for j in range(3))
        m = s >= 5
    return m.astype(np.uint8)


def _hv_like(seed: int, shape=(80, 96)) -> np.ndarray:
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:shape[0], 0:shape[1]]
    out = np.zeros(shape, np.float32)
    for _ in range(12):
        cy, cx, r = rng.uniform(0, shape[0]), rng.uniform(0, shape[1]), rng.uniform(4, 12)
        d = (xs - cx) / r
        out = np.where((ys - cy) ** 2 + (xs - cx) ** 2 < r * r, d, out)
    return (out + rng.normal(0, 0.02, shape)).astype(np.float32)


# ----------------------------------------------------------------------------------------------- scipy (always runs)
def test_morphology_restatement_equals_scipy_for_centred_kernels():
    """Odd elliptical kernels are point-symmetric, so OpenCV's erode / dilate (kernel applied un-reflected, anchor at the
    centre, border = identity element) coincide with ``scipy.ndimage.binary_erosion(border_value=1)`` /
    ``binary_dilation(border_value=0)``: the restatement must reproduce them bit for bit."""
    from scipy import ndimage

    for seed, k in ((1, 3), (2, 5), (3, 7), (4, 11), (5, 21)):
        mask = _blobs(seed)
        elem = cvref.get_structuring_element_ellipse((k, k))
        assert np.array_equal(elem, elem[::-1, ::-1])
        er = ndimage.binary_erosion(mask, structure=elem, border_value=1)
        di = ndimage.binary_dilation(mask, structure=elem, border_value=0)
        assert np.array_equal(cvref.morphology_ex(mask, "ERODE", elem).astype(bool), er)
        assert np.array_equal(cvref.morphology_ex(mask, "DILATE", elem).astype(bool), di)
        op = ndimage.binary_dilation(ndimage.binary_erosion(mask, structure=elem, border_value=1), structure=elem, border_value=0)
        assert np.array_equal(cvref.morphology_ex(mask, "OPEN", elem).astype(bool), op)


def test_remove_small_objects_restatement_against_scipy_label():
    from scipy import ndimage

    mask = _blobs(7, p=0.5).astype(bool)
    lab, n = ndimage.label(mask)
    areas = np.bincount(lab.ravel())
    for size in (1, 10, 64):
        keep = np.isin(lab, [i for i in range(1, n + 1) if areas[i] > size])
        got = skref.remove_small_objects(mask, size)
        assert np.array_equal(got, keep), size


# ----------------------------------------------------------------------------------------------- OpenCV
def test_cv2_colour_conversions_bit_exact():
    cv2 = pytest.importorskip("cv2")
    rng = np.random.default_rng(0)
    imgs = [synth.g_he(1, 128, 128, seed=3)[0], synth.g_uniform(1, 256, 256, seed=1)[0],
            rng.integers(0, 256, (512, 512, 3), dtype=np.uint8)]
    # a strided walk through the whole RGB cube (every 5th level per channel) + the cube's faces
    lv = np.arange(0, 256, 5, dtype=np.uint8)
    cube = np.stack(np.meshgrid(lv, lv, lv, indexing="ij"), -1).reshape(-1, 52, 3)
    imgs.append(np.ascontiguousarray(cube))
    for img in imgs:
        assert np.array_equal(cvref.rgb2lab_u8(img), cv2.cvtColor(img, cv2.COLOR_RGB2LAB))
        assert np.array_equal(cvref.rgb2gray_u8(img), cv2.cvtColor(img, cv2.COLOR_RGB2GRAY))
        lab = cv2.cvtColor(img, cv2.COLOR_RGB2LAB)
        assert np.array_equal(cvref.lab2rgb_u8(lab), cv2.cvtColor(lab, cv2.COLOR_LAB2RGB))
    # Lab values no RGB colour maps to (Reinhard's transform produces them): the inverse must agree there too
    lab = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    assert np.array_equal(cvref.lab2rgb_u8(lab), cv2.cvtColor(lab, cv2.COLOR_LAB2RGB))
    for c in range(3):
        m, s = cv2.meanStdDev(imgs[0][..., c].astype(np.float32))
        mm, ss = cvref.mean_std_dev(imgs[0][..., c].astype(np.float32))
        assert abs(float(m[0, 0]) - mm) < 1e-9 and abs(float(s[0, 0]) - ss) < 1e-9


def test_cv2_filters_exact():
    cv2 = pytest.importorskip("cv2")
    for seed in (1, 2):
        raw = _hv_like(seed)
        norm = cv2.normalize(raw, None, alpha=0, beta=1, norm_type=cv2.NORM_MINMAX, dtype=cv2.CV_32F)
        assert np.array_equal(cvref.normalize_minmax_to_f32(raw), norm)
        norm64 = cv2.normalize(raw.astype(np.float64), None, alpha=0, beta=1, norm_type=cv2.NORM_MINMAX, dtype=cv2.CV_32F)
        assert np.array_equal(cvref.normalize_minmax_to_f32(raw.astype(np.float64)), norm64)
        for ksize in (5, 11, 21):
            for dx, dy in ((1, 0), (0, 1)):
                exp = cv2.Sobel(norm, cv2.CV_64F, dx, dy, ksize=ksize)
                got = cvref.sobel_f64(norm, dx, dy, ksize)
                assert got.dtype == np.float64 and np.array_equal(got, exp), (ksize, dx, dy, np.abs(got - exp).max())
            kx, ky = cv2.getDerivKernels(1, 0, ksize, normalize=False, ktype=cv2.CV_64F)
            rx, ry = cvref.sobel_kernels(ksize, 1, 0)
            assert np.array_equal(rx, kx.ravel()) and np.array_equal(ry, ky.ravel())
        plane = cv2.Sobel(norm, cv2.CV_64F, 1, 0, ksize=21)
        assert np.array_equal(cvref.gaussian_blur3_f64(plane), cv2.GaussianBlur(plane, (3, 3), 0))


def test_cv2_morphology_labels_contours_bit_exact():
    cv2 = pytest.importorskip("cv2")
    for k in (1, 2, 3, 4, 5, 8, 10, 20, 21):   # even kernels included: MorphologicalMasker / _proc_ls use them
        assert np.array_equal(cvref.get_structuring_element_ellipse((k, k)), cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (k, k))), k
    ops = {"ERODE": cv2.MORPH_ERODE, "DILATE": cv2.MORPH_DILATE, "OPEN": cv2.MORPH_OPEN, "CLOSE": cv2.MORPH_CLOSE}
    for seed, k in ((1, 3), (2, 4), (3, 5), (4, 10), (5, 20)):
        mask = _blobs(seed)
        elem = cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (k, k))
        for name, op in ops.items():
            assert np.array_equal(cvref.morphology_ex(mask, name, elem), cv2.morphologyEx(mask, op, elem)), (k, name)
    for seed in (11, 12, 13):
        mask = _blobs(seed, p=0.5)
        for conn in (4, 8):
            n, lab, stats, _ = cv2.connectedComponentsWithStats(mask, connectivity=conn)
            rn, rlab, rstats, _ = cvref.connected_components_with_stats(mask, conn)
            assert rn == n and np.array_equal(rlab, lab) and np.array_equal(rstats[:, 4], stats[:, cv2.CC_STAT_AREA])
        for simple, mode in ((True, cv2.CHAIN_APPROX_SIMPLE), (False, cv2.CHAIN_APPROX_NONE)):
            real, _ = cv2.findContours(mask, cv2.RETR_TREE, mode)
            mine = cvref.find_contours(mask, simple=simple)
            assert len(mine) == len(real)
            for a, b in zip(mine, real):
                assert np.array_equal(a, b.reshape(-1, 2))
        assert np.array_equal(cvref.first_contour(mask), cv2.findContours(mask, cv2.RETR_TREE, cv2.CHAIN_APPROX_SIMPLE)[0][0].reshape(-1, 2))


# ----------------------------------------------------------------------------------------------- scikit-image
def test_skimage_primitives_bit_exact():
    pytest.importorskip("skimage")
    from scipy import ndimage
    from skimage import exposure, filters, morphology, segmentation

    rng = np.random.default_rng(5)
    grey = cvref.rgb2gray_u8(synth.g_he(1, 256, 256, seed=9)[0])
    assert skref.threshold_otsu_u8(grey) == filters.threshold_otsu(grey)
    two = np.where(rng.random((64, 64)) < 0.3, 40, 200).astype(np.uint8)
    assert skref.threshold_otsu_u8(two) == filters.threshold_otsu(two)
    img = synth.g_he(1, 64, 64, seed=2)[0]
    lo, hi = np.percentile(img, (2, 98))
    exp = exposure.rescale_intensity(img, in_range=(lo, hi), out_range=(0.0, 255.0))
    assert np.array_equal(skref.rescale_intensity(img, (lo, hi), (0.0, 255.0)), exp)
    mask = _blobs(3, p=0.5).astype(bool)
    try:
        real = morphology.remove_small_objects(mask, max_size=10)
    except TypeError:   # scikit-image < 0.26 spells it min_size (removes < min_size)
        real = morphology.remove_small_objects(mask, min_size=11)
    assert np.array_equal(skref.remove_small_objects(mask, 10), real)
    # watershed incl. plateau ties: quantised distance maps, many single-pixel markers
    for seed in (1, 2, 3):
        r = np.random.default_rng(seed)
        blobs = _blobs(seed + 20, shape=(72, 80), p=0.6).astype(bool)
        dist = -np.round(ndimage.distance_transform_edt(blobs) * (2 if seed < 3 else 1)) / 2
        markers = np.zeros(blobs.shape, np.int32)
        ys, xs = np.nonzero(blobs)
        pick = r.choice(len(ys), size=min(40, len(ys)), replace=False)
        markers[ys[pick], xs[pick]] = np.arange(1, len(pick) + 1)
        real = segmentation.watershed(dist, markers=markers, mask=blobs)
        assert np.array_equal(skref.watershed(dist, markers, blobs), real), seed


# ----------------------------------------------------------------------------------------------- torchvision
def test_torchvision_resnet50_topology_matches_the_restated_encoder():
    """The UNet golden's encoder runs on this repo's ``Bottleneck`` / ``_make_layer`` (torchvision is absent in the build
    container): where torchvision exists, the restated resnet50 must load torchvision's state dict strictly and give the
    same feature maps -- block topology, stride placement (on the 3x3: torchvision >= 0.x 'v1.5') and BN ordering."""
    tv = pytest.importorskip("torchvision")
    import torch

    from tiatoolbox_amd.models.architecture.resnet import resnet_trunk

    torch.manual_seed(0)
    real = tv.models.resnet50(weights=None).eval()
    for m in real.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    trunk = resnet_trunk("resnet50").eval()
    real_sd = {k: v for k, v in real.state_dict().items() if not k.startswith("fc.")}
    names = ["conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3", "layer4"]
    mapped = {}
    for k, v in real_sd.items():
        head, rest = k.split(".", 1)
        mapped[f"{names.index(head)}.{rest}"] = v
    trunk.load_state_dict(mapped, strict=True)
    x = torch.randn(2, 3, 96, 96)
    with torch.no_grad():
        a = trunk(x)
        b = real.layer4(real.layer3(real.layer2(real.layer1(real.maxpool(real.relu(real.bn1(real.conv1(x))))))))
    assert torch.allclose(a, b, atol=1e-5), float((a - b).abs().max())
