"""Reinhard normaliser + 8-bit Lab conversions: oracle vs the real reference (CPU), HIP vs oracle (GPU)."""

from __future__ import annotations

from pathlib import Path

import numpy as np
import pytest

from oracle import cvref
from oracle import stain as ostain
from tiatoolbox_amd.utils import synth

GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD / "reinhard_golden.npz")


@pytest.fixture(scope="module")
def images():
    g = np.load(GOLD / "stain_golden.npz")
    return {"real": g["real_crops"], "he": synth.g_he(3, 96, 96, seed=int(g["he_seed"]))}


def test_oracle_matches_real_reference(gold, target_image, images):
    norm = ostain.get_normalizer("reinhard")
    norm.fit(target_image.copy())
    np.testing.assert_allclose(norm.target_means, gold["target_means"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(norm.target_stds, gold["target_stds"], rtol=0, atol=1e-12)
    for key, imgs in images.items():
        assert np.array_equal(np.stack([norm.transform(c.copy()) for c in imgs]), gold[key])


def test_lab_round_trip_is_close():
    """Sanity of the (unpinned) 8-bit Lab restatement: RGB -> Lab -> RGB stays within quantisation error."""
    rng = np.random.default_rng(0)
    grey = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)
    assert np.abs(cvref.lab2rgb_u8(cvref.rgb2lab_u8(grey)).astype(int) - grey).max() <= 1
    rgb = rng.integers(40, 216, (64, 64, 3), dtype=np.uint8)
    back = cvref.lab2rgb_u8(cvref.rgb2lab_u8(rgb))
    assert np.abs(back.astype(int) - rgb).mean() < 1.0


@pytest.mark.gpu
def test_hip_lab_conversions_bit_exact():
    import torch

    from tiatoolbox_amd.tools import reinhard

    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (3, 50, 70, 3), dtype=np.uint8)
    lab = reinhard.lab_convert(torch.from_numpy(rgb).cuda(), 0).cpu().numpy()
    assert np.array_equal(lab, cvref.rgb2lab_u8(rgb))
    every_lab = rng.integers(0, 256, (4, 64, 64, 3), dtype=np.uint8)
    back = reinhard.lab_convert(torch.from_numpy(every_lab).cuda(), 1).cpu().numpy()
    assert np.array_equal(back, cvref.lab2rgb_u8(every_lab))


@pytest.mark.gpu
def test_hip_reinhard_matches_reference_goldens(gold, target_image, images):
    from tiatoolbox_amd.tools.stainnorm import get_normalizer

    norm = get_normalizer("reinhard")
    norm.fit(target_image)
    np.testing.assert_allclose(norm.target_means, gold["target_means"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(norm.target_stds, gold["target_stds"], rtol=0, atol=1e-9)
    means, stds = zip(*[norm.get_mean_std(c) for c in images["real"]])
    np.testing.assert_allclose(np.array(means), gold["real_means"], atol=1e-9)
    np.testing.assert_allclose(np.array(stds), gold["real_stds"], atol=1e-9)
    for key, imgs in images.items():
        out = norm.transform(imgs)           # batched
        assert out.dtype == np.uint8 and np.array_equal(out, gold[key]), key
        assert np.array_equal(norm.transform(imgs[0]), gold[key][0])   # single image
    # lab_split / merge_back API
    c1, c2, c3 = norm.lab_split(images["real"][0])
    e1, e2, e3 = ostain.ReinhardNormalizer.lab_split(images["real"][0])
    assert np.array_equal(c1, e1) and np.array_equal(c2, e2) and np.array_equal(c3, e3)
    assert np.array_equal(norm.merge_back(c1.copy(), c2.copy(), c3.copy()),
                          ostain.ReinhardNormalizer.merge_back(e1.copy(), e2.copy(), e3.copy()))


@pytest.mark.gpu
def test_hip_reinhard_device_luts_match_host_arithmetic():
    """tia_reinhard_luts (statistics + float32 LUT chain on the device) == the NumPy evaluation of the same
    expressions (``ReinhardNormalizer._mean_std`` / ``_luts``), bit for bit, incl. the zero-std error."""
    import ctypes as C

    import torch

    from tiatoolbox_amd import _lib
    from tiatoolbox_amd.tools import reinhard as rh

    rng = np.random.default_rng(5)
    n = 300
    hist = np.zeros((n, 3, 256), np.int64)
    for i in range(n):
        for c in range(3):
            lo, hi = sorted(rng.integers(0, 256, 2))
            lo, hi = min(lo, 253), min(max(hi, lo + 2), 256)
            vals = rng.integers(lo, hi, rng.integers(50, 60000))
            hist[i, c] = np.bincount(vals, minlength=256)
    norm = rh.ReinhardNormalizer()
    norm.target_means, norm.target_stds = (63.123456789, 7.25, -4.875), (11.0625, 5.3, 6.789)
    means, stds = norm._mean_std(hist)  # noqa: SLF001
    exp = norm._luts(means, stds)       # noqa: SLF001
    dev = torch.device("cuda")
    hist_d = torch.from_numpy(hist.astype(np.int32)).to(dev)
    luts = torch.empty((n, 3, 256), dtype=torch.uint8, device=dev)
    ms = torch.empty((n, 6), dtype=torch.float64, device=dev)
    flags = torch.zeros(n, dtype=torch.int32, device=dev)
    tm, ts = (C.c_double * 3)(*norm.target_means), (C.c_double * 3)(*norm.target_stds)
    rc = _lib.load().tia_reinhard_luts(hist_d.data_ptr(), n, rh._channel_values_device(dev).data_ptr(), tm, ts,  # noqa: SLF001
                                       luts.data_ptr(), ms.data_ptr(), flags.data_ptr(), _lib.current_stream())
    _lib.check(rc, "tia_reinhard_luts")
    got_ms = ms.cpu().numpy()
    assert np.array_equal(got_ms[:, :3], means) and np.array_equal(got_ms[:, 3:], stds)
    assert np.array_equal(luts.cpu().numpy(), exp)
    assert int(flags.sum()) == 0
    flat = np.full((4, 4, 3), 200, np.uint8)
    norm.fit(np.random.default_rng(0).integers(0, 255, (16, 16, 3), dtype=np.uint8))
    with pytest.raises(ZeroDivisionError):
        norm.transform(flat)


@pytest.mark.gpu
def test_hip_lab_conversions_exhaustive():
    """EVERY 24-bit RGB value through RGB2LAB and every Lab triple through LAB2RGB, against the oracle's integer arithmetic: the float32
    forward path (exact sums below 2^24, round-half-up through v_cvt_pk_u8_f32) and the 24-bit-multiply inverse are bit-exact everywhere,
    not only on samples.  16.7 M pixels = 16384 wave steps of the 16-byte path."""
    import torch

    from tiatoolbox_amd.tools import reinhard

    v = np.arange(1 << 24, dtype=np.uint32)
    allpx = np.stack([v & 255, (v >> 8) & 255, v >> 16], axis=-1).astype(np.uint8).reshape(4096, 4096, 3)
    dev = torch.from_numpy(allpx).cuda()
    lab = reinhard.lab_convert(dev, 0).cpu().numpy()
    back = reinhard.lab_convert(dev, 1).cpu().numpy()
    for s in range(0, 4096, 512):   # the oracle in slabs (memory)
        assert np.array_equal(lab[s:s + 512], cvref.rgb2lab_u8(allpx[s:s + 512])), s
        assert np.array_equal(back[s:s + 512], cvref.lab2rgb_u8(allpx[s:s + 512])), s


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(5, 64, 64), (3, 32, 32), (2, 224, 224), (1, 1000, 1000), (2, 37, 53), (1, 1024, 1536)])
def test_hip_reinhard_paths_match_oracle(shape, target_image):
    """transform / statistics on every route -- the one-launch fused kernel (h*w a multiple of 1024), the three-launch form with the
    16-byte kernels (a large image split over many workgroups) and its scalar form (ragged sizes) -- against the oracle, bit for bit:
    noise, a flat white region (every lane on one histogram counter), near-black pixels (the cube's linear branch)."""
    import torch

    from tiatoolbox_amd.tools.stainnorm import get_normalizer

    n, h, w = shape
    rng = np.random.default_rng(h * 7 + w)
    imgs = synth.g_he(n, h, w, seed=h + w)
    imgs[0, : h // 3] = 255
    imgs[0, h // 3: h // 3 + max(1, h // 8)] = rng.integers(0, 12, (max(1, h // 8), w, 3), dtype=np.uint8)
    imgs[-1, :, : w // 2] = rng.integers(0, 256, (h, w // 2, 3), dtype=np.uint8)
    norm, ref = get_normalizer("reinhard"), ostain.get_normalizer("reinhard")
    norm.fit(target_image)
    ref.fit(target_image.copy())
    got = norm.transform(torch.from_numpy(imgs).cuda()).cpu().numpy()
    for i in range(n):
        assert np.array_equal(got[i], ref.transform(imgs[i].copy())), (shape, i)
    ms = norm.lab_statistics(torch.from_numpy(imgs).cuda()).cpu().numpy()
    for i in range(n):
        mean, std = ref.get_mean_std(imgs[i].copy())
        np.testing.assert_allclose(ms[i, :3], mean, rtol=0, atol=1e-9)
        np.testing.assert_allclose(ms[i, 3:], std, rtol=0, atol=1e-9)
        m2, s2 = norm.get_mean_std(imgs[i])
        assert np.array_equal(ms[i], np.concatenate([m2, s2]))   # device moments == the host arithmetic on the device's counts
