"""Tissue maskers: oracle vs the reference's known answers / real outputs (CPU), HIP vs oracle (GPU)."""

from __future__ import annotations

from pathlib import Path

import numpy as np
import pytest

from oracle import tissuemask as omask
from tiatoolbox_amd.utils import synth

GOLD = Path(__file__).parent / "golden"
MORPH_CASES = (("k1", {"kernel_size": 1, "min_region_size": 6}), ("p125", {"power": 1.25}),
               ("k5", {"kernel_size": 5}), ("mpp4", {"mpp": (4.0, 7.0)}))


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD / "mask_golden.npz")


@pytest.fixture(scope="module")
def images():
    return {"real": np.load(GOLD / "stain_golden.npz")["real_crops"], "he": synth.g_he(2, 160, 200, seed=31)}


def _known_answer_image():
    """Reference tests/test_tissuemask.py:186-212 (0 = foreground, 1 = background)."""
    img = np.ones((10, 10))
    img[1:4, 1:4] = 0
    img[1, 5:10] = 0
    img[8, 8] = 0
    expected = np.zeros((10, 10))
    expected[1:4, 1:4] = 1
    return img, expected


def _check_contracts(mod):
    with pytest.raises(SyntaxError, match="Fit must be called before transform"):
        mod.OtsuTissueMasker().transform(np.zeros((1, 4, 4, 3), np.uint8))
    with pytest.raises(SyntaxError, match="Fit must be called before transform"):
        mod.MorphologicalMasker().transform(np.zeros((1, 4, 4, 3), np.uint8))
    with pytest.raises(ValueError, match="Expected 4 dimensional input shape"):
        mod.OtsuTissueMasker().fit(np.zeros((4, 4, 3), np.uint8))
    with pytest.raises(ValueError, match="Only one of mpp, power, kernel_size can be given"):
        mod.MorphologicalMasker(mpp=0.25, power=40)
    assert mod.MorphologicalMasker(kernel_size=None).kernel_size == (1, 1)


def test_oracle_contracts_and_known_answer():
    _check_contracts(omask)
    img, expected = _known_answer_image()
    out = omask.MorphologicalMasker(kernel_size=1, min_region_size=6).fit_transform([img[..., np.newaxis]])
    assert np.sum(out[0]) == 9 and np.all(out[0] == expected)


def test_oracle_matches_real_reference(gold, images):
    for name, imgs in images.items():
        m = omask.OtsuTissueMasker()
        assert np.array_equal(m.fit_transform(imgs), gold[f"otsu_{name}"])
        assert m.threshold == gold[f"otsu_thr_{name}"]
        for tag, kw in MORPH_CASES:
            mm = omask.MorphologicalMasker(**kw)
            assert np.array_equal(mm.kernel, gold[f"morph_{tag}_kernel"])
            assert mm.min_region_size == int(gold[f"morph_{tag}_minreg"])
            assert np.array_equal(mm.fit_transform(imgs), gold[f"morph_{tag}_{name}"])


@pytest.mark.gpu
def test_hip_maskers_bit_exact(gold, images):
    from tiatoolbox_amd.tools import tissuemask as hmask

    _check_contracts(hmask)
    img, expected = _known_answer_image()
    out = hmask.MorphologicalMasker(kernel_size=1, min_region_size=6).fit_transform([img[..., np.newaxis]])
    assert out.dtype == bool and np.sum(out[0]) == 9 and np.all(out[0] == expected)
    for name, imgs in images.items():
        m = hmask.OtsuTissueMasker()
        got = m.fit_transform(imgs)
        assert got.dtype == bool and np.array_equal(got, gold[f"otsu_{name}"])
        assert m.threshold == gold[f"otsu_thr_{name}"]
        for tag, kw in MORPH_CASES:
            mm = hmask.MorphologicalMasker(**kw)
            assert np.array_equal(mm.kernel, gold[f"morph_{tag}_kernel"])
            assert np.array_equal(mm.fit_transform(imgs), gold[f"morph_{tag}_{name}"]), (name, tag)


@pytest.mark.gpu
@pytest.mark.parametrize("conn", [4, 8])
def test_ccl_matches_scipy_label(conn):
    """Label numbering = raster order of first pixel, exactly scipy.ndimage.label; random + adversarial masks."""
    import torch
    from scipy import ndimage

    from tiatoolbox_amd.tools import _img_device as img

    rng = np.random.default_rng(7)
    planes = [rng.random((97, 131)) < p for p in (0.3, 0.5, 0.62, 0.9)]
    spiral = np.zeros((97, 131), bool)
    spiral[::2, :] = True
    spiral[1::4, -1] = True
    spiral[3::4, 0] = True   # one long serpentine component
    planes += [spiral, np.zeros((97, 131), bool), np.ones((97, 131), bool)]
    mask = np.stack(planes)
    labels, count = img.ccl_label(torch.from_numpy(mask).cuda().to(torch.uint8), connectivity=conn)
    structure = np.ones((3, 3), int) if conn == 8 else None
    for i, m in enumerate(mask):
        exp, n = ndimage.label(m, structure=structure)
        assert int(count[i]) == n
        assert np.array_equal(labels[i].cpu().numpy(), exp)
    # area filter == skimage remove_small_objects(max_size=9) semantics (no relabel)
    from oracle import skref

    filt = img.label_area_filter(labels.clone(), 10).cpu().numpy()
    for i in range(len(mask)):
        assert np.array_equal(filt[i], skref.remove_small_objects_labels(labels[i].cpu().numpy(), 9))
    # fill holes == scipy binary_fill_holes
    filled = img.fill_holes(torch.from_numpy(mask).cuda().to(torch.uint8)).cpu().numpy()
    for i, m in enumerate(mask):
        assert np.array_equal(filled[i].astype(bool), ndimage.binary_fill_holes(m))


@pytest.mark.gpu
def test_large_plane_ccl_properties():
    """1024x1024 tile (HoVer-Net WSI tile size): label count and areas agree with scipy; idempotent."""
    import torch
    from scipy import ndimage

    from tiatoolbox_amd.tools import _img_device as img

    rng = np.random.default_rng(3)
    m = ndimage.binary_dilation(rng.random((1024, 1024)) < 0.02, iterations=2)
    labels, count = img.ccl_label(torch.from_numpy(m).cuda().to(torch.uint8), connectivity=4)
    exp, n = ndimage.label(m)
    assert int(count[0]) == n and np.array_equal(labels[0].cpu().numpy(), exp)
    relabel, count2 = img.ccl_label((labels > 0).to(torch.uint8), connectivity=4)
    assert torch.equal(relabel, labels) and int(count2[0]) == n
