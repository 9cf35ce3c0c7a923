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


@pytest.mark.gpu
def test_device_otsu_threshold_matches_numpy_arithmetic():
    """tia_otsu_threshold_u32 == the host's scikit-image arithmetic on the occupied range, for adversarial histograms."""
    import torch

    from tiatoolbox_amd.tools import _img_device as img
    from tiatoolbox_amd.tools.tissuemask import _otsu_from_counts

    rng = np.random.default_rng(5)
    cases = [rng.integers(0, 1000, 256), rng.integers(0, 2, 256) * rng.integers(1, 1 << 20, 256), np.zeros(256, np.int64), np.zeros(256, np.int64),
             np.zeros(256, np.int64), (rng.random(256) ** 8 * 4e8).astype(np.int64), np.full(256, 7), np.zeros(256, np.int64)]
    cases[2][17] = 5                                     # one occupied bin
    cases[3][[3, 250]] = (9, 9)                          # two bins, far apart (flat maximum: first index wins)
    cases[4][[100, 101]] = (1, 4_000_000_000)            # two adjacent bins, counts near 2^32
    cases[7][[0, 128, 255]] = (10, 1, 10)                # symmetric: tie between the two halves
    for counts in cases:
        counts = np.asarray(counts, dtype=np.int64)
        nz = np.flatnonzero(counts)
        if nz.size == 1:
            exp = int(nz[0])
        else:
            lo, hi = int(nz[0]), int(nz[-1])
            exp = int(_otsu_from_counts(counts[lo:hi + 1], np.arange(lo, hi + 1, dtype=np.float64)))
        dev = torch.from_numpy(counts.astype(np.uint32).view(np.int32)).cuda()
        out = img.otsu_threshold(dev).cpu().numpy()
        assert int(out[0]) == exp and int(out[1]) == nz.size, (counts[nz][:8], out, exp)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 64, 64), (1, 37, 53), (2, 128, 96), (1, 1, 5), (5, 160, 200)])
def test_fused_grey_histogram_and_wide_threshold(shape):
    """One-pass grey + histogram == bincount of the oracle's grey image; the 16-byte threshold kernel == grey < thr: sizes that are
    and are not multiples of the 1024-pixel wave step, aligned and unaligned bases, an image of one grey level."""
    import torch

    from tiatoolbox_amd.tools import _img_device as img

    rng = np.random.default_rng(shape[1])
    n, h, w = shape
    rgb = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    rgb[0, : h // 2] = 200                                            # a flat half: every lane of a wave on one counter
    grey = np.stack([omask._grey(i) for i in rgb])                     # noqa: SLF001
    for off in (0, 3):                                                # offset 3 bytes: the unaligned (scalar) path
        flat = torch.zeros(rgb.size + 16, dtype=torch.uint8, device="cuda")
        view = flat[off:off + rgb.size].view(n, h, w, 3)
        view.copy_(torch.from_numpy(rgb))
        counts = img.gray_hist(view, channels=3).cpu().numpy()
        assert np.array_equal(counts, np.bincount(grey.ravel(), minlength=256))
        assert np.array_equal(img.gray_hist(torch.from_numpy(grey).cuda(), channels=1).cpu().numpy(), counts)
        for thr in (0, 97, 200, 201, 256):
            got = img.threshold_lt(view, thr, is_rgb=True).cpu().numpy()
            assert np.array_equal(got, (grey < thr).astype(np.uint8)), (off, thr)
        thr_dev = torch.tensor([131, 0], dtype=torch.int32, device="cuda")
        assert np.array_equal(img.threshold_lt(view, thr_dev, is_rgb=True).cpu().numpy(), (grey < 131).astype(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize(("kw", "shape"), [({"power": 1.25}, (2, 700, 900)), ({"kernel_size": 5}, (1, 300, 1100)),
                                            ({"kernel_size": (7, 3), "min_region_size": 30}, (1, 513, 257)),
                                            ({"kernel_size": 1, "min_region_size": 0}, (1, 130, 260)),
                                            ({"mpp": 2.0}, (1, 400, 400))])
def test_one_launch_morphological_masker_matches_oracle(kw, shape):
    """The tile kernel (threshold + small-region removal + dilation in LDS, halo = element reach + min_region_size - 1) against the
    oracle on images of several tiles: components that cross tile borders, thin lines longer than the halo but smaller than the
    area bound and the other way round, specks at every density, image borders; the last case (16 x 16 element: halo > 40) is the
    multi-launch fallback."""
    import torch

    from tiatoolbox_amd.tools import tissuemask as hmask

    n, h, w = shape
    rng = np.random.default_rng(h + w)
    grey = np.full((n, h, w), 230, np.uint8)
    for i in range(n):
        for p in (0.002, 0.02, 0.2):                      # specks: isolated pixels .. touching clusters
            y0, x0 = rng.integers(0, h // 2), rng.integers(0, w // 2)
            blk = rng.random((h // 2, w // 2)) < p
            grey[i, y0:y0 + h // 2, x0:x0 + w // 2][blk] = 40
        for _ in range(40):                               # lines: horizontal, vertical, diagonal, 2 .. 60 pixels
            ln, y, x = int(rng.integers(2, 60)), int(rng.integers(0, h)), int(rng.integers(0, w))
            dy, dx = [(0, 1), (1, 0), (1, 1), (1, -1)][int(rng.integers(0, 4))]
            for t in range(ln):
                yy, xx = y + t * dy, x + t * dx
                if 0 <= yy < h and 0 <= xx < w:
                    grey[i, yy, xx] = 40
        grey[i, :3, : w // 3] = 40                        # on the image border
        grey[i, h - 1, w - 5:] = 40
        grey[i, 100:104, 200:230] = 40
    rgb = np.repeat(grey[..., None], 3, axis=-1)
    rgb[..., 1] = np.clip(rgb[..., 1].astype(int) + rng.integers(-3, 4, grey.shape), 0, 255).astype(np.uint8)
    ref = omask.MorphologicalMasker(**kw)
    exp = ref.fit_transform(rgb)
    got_m = hmask.MorphologicalMasker(**kw)
    got = got_m.fit_transform(torch.from_numpy(rgb).cuda())
    assert got.dtype == torch.bool and got_m.threshold == ref.threshold
    assert np.array_equal(got.cpu().numpy(), exp), (kw, int((got.cpu().numpy() != exp).sum()))
    one = hmask.MorphologicalMasker(**kw)                  # grey (single-channel) input: same masks from the grey plane
    g3 = np.stack([omask._grey(i) for i in rgb])           # noqa: SLF001
    one.fit(torch.from_numpy(rgb).cuda())
    assert np.array_equal(one.transform(torch.from_numpy(g3[..., None]).cuda()).cpu().numpy(), exp)


@pytest.mark.gpu
def test_one_launch_otsu_fit_is_race_free():
    """The fit kernel's LAST workgroup computes the threshold from counts the other workgroups delivered as device-scope atomics
    (ordering by acknowledgement, no release fence).  150 fits over images of changing size / content: the threshold and the occupied
    bin count always equal those of the two-launch form (histogram, then the threshold kernel) on the same pixels."""
    import torch

    from tiatoolbox_amd.tools import _img_device as img

    rng = np.random.default_rng(11)
    for it in range(150):
        n, h, w = int(rng.integers(1, 5)), int(rng.integers(40, 900)), int(rng.integers(40, 900))
        lo, hi = sorted(int(v) for v in rng.integers(0, 256, 2))
        x = torch.from_numpy(rng.integers(lo, hi + 1, (n, h, w, 3), dtype=np.uint8)).cuda()
        if it % 7 == 0:
            x[0, : h // 2] = 255
        got = img.otsu_fit(x, channels=3)
        exp = img.otsu_threshold(img.gray_hist(x, channels=3))
        assert torch.equal(got, exp), (it, n, h, w, got.tolist(), exp.tolist())
