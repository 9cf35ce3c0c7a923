"""Parity at the FULL sizes of BASELINE configs[2] and configs[4], checked on samples against the oracle.

* configs[2]: one 20 000 x 20 000 slide through ``SemanticSegmentor`` WSI mode (1024-in / 512-out patches, stride 450,
  tissue mask) with a deterministic stub head; 20 windows of 512 x 512 of the stitched probabilities / predictions
  against ``oracle.semantic`` -- the reference's ``merge_horizontal`` / ``merge_vertical_chunkwise`` arithmetic
  (``/root/reference/tiatoolbox/models/engine/semantic_segmentor.py:1141-1263,1398-1534``) -- bit for bit.
* configs[4]: one 8192 x 256 x 256 Vahadane launch sequence; 16 patches spread over the batch against scikit-learn's
  ``DictionaryLearning`` driven as the reference drives it (``tools/stainextract.py:281-322``), <= 1e-6, and the
  ``StainAugmentor`` output of those patches within 1 LSB.
"""

from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import semantic as osem
from oracle import stain as ostain

SIDE, PIN, POUT, STRIDE, NCH = 20000, 1024, 512, 450, 5


def _stub_values(patch_u8, xp):
    """Deterministic head output of one batch of input patches ``[n, 1024, 1024, 3]`` -> ``[n, 512, 512, 5]`` float32 in
    [0, 1): 24-bit fractions built from the centre crop's bytes and the position inside the patch, so overlapping patches
    disagree and their float32 sums round (the order of the additions is visible in the last bit).  ``xp`` = torch / numpy:
    integer arithmetic + one exact int -> float conversion + one exact scaling, identical on both."""
    off = (PIN - POUT) // 2
    if xp is torch:
        t = patch_u8[:, off:off + POUT, off:off + POUT, :].to(torch.int32)
        yy = torch.arange(POUT, device=t.device, dtype=torch.int32).view(1, POUT, 1)
        xx = torch.arange(POUT, device=t.device, dtype=torch.int32).view(1, 1, POUT)
        chans = []
        for c in range(NCH):
            b0 = (t[..., c % 3] * (c + 3) + yy * 7 + xx * 13) & 255
            b1 = (t[..., (c + 1) % 3] * 5 + xx * 3 + c) & 255
            b2 = (t[..., (c + 2) % 3] + yy) & 255
            chans.append((b0 * 65536 + b1 * 256 + b2).to(torch.float32) * (2.0 ** -24))
        return torch.stack(chans, dim=-1)
    t = patch_u8[:, off:off + POUT, off:off + POUT, :].astype(np.int32)
    yy = np.arange(POUT, dtype=np.int32).reshape(1, POUT, 1)
    xx = np.arange(POUT, dtype=np.int32).reshape(1, 1, POUT)
    chans = []
    for c in range(NCH):
        b0 = (t[..., c % 3] * (c + 3) + yy * 7 + xx * 13) & 255
        b1 = (t[..., (c + 1) % 3] * 5 + xx * 3 + c) & 255
        b2 = (t[..., (c + 2) % 3] + yy) & 255
        chans.append((b0 * 65536 + b1 * 256 + b2).astype(np.float32) * np.float32(2.0 ** -24))
    return np.stack(chans, axis=-1)


class _StubHead(torch.nn.Module):
    """Stands in for the UNet: same ``infer_batch`` contract (``architecture/unet.py:374-418``), deterministic output."""

    def __init__(self) -> None:
        super().__init__()
        self.dummy = torch.nn.Parameter(torch.zeros(1))

    preproc_func = staticmethod(lambda img: img)
    postproc_func = staticmethod(lambda img: img)

    @staticmethod
    def infer_batch(model, batch_data, *, device):  # noqa: ARG004
        return _stub_values(batch_data, torch)


@pytest.mark.gpu
def test_semantic_wsi_stitching_at_full_size_sampled_against_oracle():
    from tiatoolbox_amd.models.engine.io_config import IOSegmentorConfig
    from tiatoolbox_amd.models.engine.semantic_segmentor import SemanticSegmentor
    from tiatoolbox_amd.wsicore import ArrayWSIReader

    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(20)
    # slide: dark noisy "tissue" blocks on a bright background; the right and bottom margins and two gutters are empty, so
    # the Otsu mask drops whole patches (rows with different numbers of kept patches, fully empty rows)
    slide = torch.randint(232, 250, (SIDE, SIDE, 3), dtype=torch.uint8, device=dev, generator=g)
    tile = torch.randint(20, 200, (4096, 4096, 3), dtype=torch.uint8, device=dev, generator=g)
    for y0 in (600, 5500, 14800):
        for x0 in (300, 4900, 9800, 15200):
            h, w = min(4096, SIDE - 900 - y0), min(4096, SIDE - 700 - x0)
            slide[y0:y0 + h, x0:x0 + w] = tile[:h, :w]
    reader = ArrayWSIReader(slide, mpp=0.25, power=40.0)
    cfg = IOSegmentorConfig(input_resolutions=[{"units": "mpp", "resolution": 0.25}],
                            output_resolutions=[{"units": "mpp", "resolution": 0.25}], patch_input_shape=[PIN, PIN],
                            patch_output_shape=[POUT, POUT], stride_shape=[STRIDE, STRIDE],
                            save_resolution={"units": "mpp", "resolution": 0.25})
    eng = SemanticSegmentor(_StubHead(), batch_size=8, device="cuda", verbose=False)
    eng._ioconfig = eng.ioconfig = cfg  # noqa: SLF001
    mask_reader = reader.tissue_mask(resolution=1.25, units="power")
    out = eng.infer_wsi(reader, mask_reader, return_probabilities=True)
    pred, probs = out["predictions"], out["probabilities"]
    assert pred.shape == (SIDE, SIDE) and probs.shape == (SIDE, SIDE, NCH)
    in_b, out_b, keep = eng.get_coordinates(reader, mask_reader)
    n_side = -(-SIDE // STRIDE)
    assert len(out_b) == n_side * n_side == 2025 and 200 < keep.sum() < len(keep)
    row_ys = np.unique(out_b[:, 1])
    kept_per_row = np.array([(keep & (out_b[:, 1] == y)).sum() for y in row_ys])
    assert (kept_per_row == 0).any() and len(np.unique(kept_per_row)) > 2  # ragged rows, some of them empty

    rng = np.random.default_rng(8)
    kept_xy = out_b[keep][:, :2]
    windows = [(0, 0), (SIDE - POUT, SIDE - POUT), (0, SIDE - POUT), (SIDE - POUT, 0)]
    windows += [(int(y) + int(rng.integers(-300, 300)), int(x) + int(rng.integers(-300, 300)))
                for x, y in kept_xy[rng.choice(len(kept_xy), 12, replace=False)]]
    windows += [(int(rng.integers(0, SIDE - POUT)), int(rng.integers(0, SIDE - POUT))) for _ in range(4)]
    nonzero_windows = 0
    for wy, wx in windows:
        wy, wx = min(max(wy, 0), SIDE - POUT), min(max(wx, 0), SIDE - POUT)
        rows = np.flatnonzero((row_ys < wy + POUT) & (row_ys + POUT > wy))
        sel = np.flatnonzero(np.isin(out_b[:, 1], row_ys[rows]) & (out_b[:, 0] < wx + POUT) & (out_b[:, 2] > wx))
        sy0, sx0 = int(row_ys[rows[0]]), int(out_b[sel, 0].min())
        sub_h = min(int(row_ys[rows[-1]]) + POUT, SIDE) - sy0
        sub_w = min(int(out_b[sel, 2].max()), SIDE) - sx0
        patches = reader.read_bounds_batch(in_b[sel]).cpu().numpy()  # 255 outside the slide, like WSIPatchDataset
        blocks = _stub_values(patches, np)
        blocks[~keep[sel]] = 0.0  # patches the tissue mask dropped are never inferred: all-zero blocks are skipped (:1178-1179)
        locs = out_b[sel].copy()
        locs[:, [0, 2]] -= sx0
        locs[:, [1, 3]] -= sy0
        exp = osem.merge_wsi(blocks, locs, (sub_h, sub_w))
        ey, ex = wy - sy0, wx - sx0
        exp_win = exp[ey:ey + POUT, ex:ex + POUT]
        got_p = probs[wy:wy + POUT, wx:wx + POUT].cpu().numpy()
        got_y = pred[wy:wy + POUT, wx:wx + POUT].cpu().numpy()
        assert exp_win.shape == got_p.shape == (POUT, POUT, NCH)
        assert np.array_equal(got_p, exp_win), (wy, wx, np.abs(got_p - exp_win).max())
        assert np.array_equal(got_y, exp_win.argmax(-1).astype(np.uint8)), (wy, wx)
        nonzero_windows += bool(exp_win.any())
    assert nonzero_windows >= 14
    # whole-map sanity at full size: rows / columns no kept patch reaches stay zero, the covered area does not
    covered = torch.zeros((SIDE,), dtype=torch.bool, device=dev)
    for y in np.unique(out_b[keep][:, 1]):
        covered[int(y):min(int(y) + POUT, SIDE)] = True
    assert not probs[~covered].any() and bool(probs[covered].any())

    # the same slide as if it did not fit the device (reference: zarr spill above `memory_threshold`,
    # semantic_segmentor.py:552-583,1693-1730): at most TWO patch rows of canvas on the device, finished rows streamed to
    # page-locked host memory on a copy stream -- the maps must be the resident run's, bit for bit
    eng.device_band_rows = 2
    out_s = eng.infer_wsi(reader, mask_reader, return_probabilities=True)
    assert eng.last_band_streamed and not out_s["predictions"].is_cuda and not out_s["probabilities"].is_cuda
    assert out_s["predictions"].shape == pred.shape and out_s["probabilities"].shape == probs.shape
    for y in range(0, SIDE, 2000):
        assert torch.equal(out_s["predictions"][y:y + 2000].to(dev), pred[y:y + 2000]), y
        assert torch.equal(out_s["probabilities"][y:y + 2000].to(dev), probs[y:y + 2000]), y


@pytest.mark.gpu
def test_streamed_canvas_band_peak_memory_is_independent_of_slide_height():
    """``CanvasBand`` in streamed mode: the device holds K chunks of ``oh`` canvas rows + the two patch rows being merged, so the
    peak device memory of ``infer_wsi`` (beyond the slide itself) does not grow with the slide's height; the resident mode's does,
    by the size of the maps.  The automatic switch (`memory_threshold`) picks the streamed mode when the maps would not fit."""
    from tiatoolbox_amd.models.engine.io_config import IOSegmentorConfig
    from tiatoolbox_amd.models.engine.semantic_segmentor import SemanticSegmentor
    from tiatoolbox_amd.wsicore import ArrayWSIReader

    dev = torch.device("cuda")
    cfg = IOSegmentorConfig(input_resolutions=[{"units": "mpp", "resolution": 0.25}],
                            output_resolutions=[{"units": "mpp", "resolution": 0.25}], patch_input_shape=[PIN, PIN],
                            patch_output_shape=[POUT, POUT], stride_shape=[STRIDE, STRIDE],
                            save_resolution={"units": "mpp", "resolution": 0.25})
    width = 4096

    def peak(height: int, rows):
        g = torch.Generator(device="cuda").manual_seed(3)
        reader = ArrayWSIReader(torch.randint(20, 200, (height, width, 3), dtype=torch.uint8, device=dev, generator=g), mpp=0.25, power=40.0)
        eng = SemanticSegmentor(_StubHead(), batch_size=4, device="cuda", verbose=False)
        eng._ioconfig = eng.ioconfig = cfg  # noqa: SLF001
        eng.device_band_rows = rows
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        base = torch.cuda.memory_allocated()
        torch.cuda.reset_peak_memory_stats()
        out = eng.infer_wsi(reader, None, return_probabilities=True)
        torch.cuda.synchronize()
        grown = torch.cuda.max_memory_allocated() - base
        assert out["predictions"].shape == (height, width) and eng.last_band_streamed == (rows is not None)
        return grown, out

    maps_bytes = lambda h: h * width * (1 + 4 * NCH)  # noqa: E731
    s_short, out_short = peak(3000, 2)
    s_tall, out_tall = peak(9000, 2)
    r_short, res_short = peak(3000, None)
    r_tall, _ = peak(9000, None)
    assert abs(s_tall - s_short) < 8 << 20, (s_short, s_tall)                      # streamed: flat in the height
    assert r_tall - r_short > 0.9 * (maps_bytes(9000) - maps_bytes(3000))          # resident: grows by the maps
    assert s_tall < r_tall - 0.9 * maps_bytes(9000) + (64 << 20)
    assert torch.equal(out_short["predictions"].to(dev), res_short["predictions"])
    assert torch.equal(out_short["probabilities"].to(dev), res_short["probabilities"])
    # the automatic switch: with a threshold of (nearly) zero per cent of the free memory every slide is "too large"
    reader = ArrayWSIReader(torch.randint(20, 200, (3000, width, 3), dtype=torch.uint8, device=dev), mpp=0.25, power=40.0)
    eng = SemanticSegmentor(_StubHead(), batch_size=4, device="cuda", verbose=False)
    eng._ioconfig = eng.ioconfig = cfg  # noqa: SLF001
    eng.memory_threshold = 1e-9
    eng.infer_wsi(reader, None, return_probabilities=False)
    assert eng.last_band_streamed
    eng.memory_threshold = 80
    eng.infer_wsi(reader, None, return_probabilities=False)
    assert not eng.last_band_streamed


@pytest.mark.gpu
def test_vahadane_and_augmentor_at_full_batch_size_sampled_against_sklearn():
    from tiatoolbox_amd import _lib
    from tiatoolbox_amd.tools import _stain_device as dev
    from tiatoolbox_amd.tools.stainaugment import StainAugmentor
    from tiatoolbox_amd.tools.stainextract import VahadaneExtractor
    from tiatoolbox_amd.utils import synth

    n, hw = 8192, 256
    uniq = synth.g_he(256, hw, hw, seed=91)
    batch = torch.from_numpy(uniq).cuda().repeat(n // 256, 1, 1, 1)
    pick = np.linspace(0, n - 1, 16).astype(int)
    for i in pick:  # make the sampled patches unique in content: roll each one by its index
        batch[i] = torch.roll(batch[i], shifts=(int(i) % hw, int(i * 7) % hw), dims=(0, 1))
    ex = VahadaneExtractor()
    stats = dev.stain_stats(batch, ex.stats_params())  # the 8192-patch launch sequence (dictionary scratch in chunks)
    got = stats[:, _lib.ST_STAIN:_lib.ST_STAIN + 6].reshape(n, 2, 3)
    flags = stats[:, _lib.ST_FLAGS].to(torch.int64)
    assert not bool(flags.any())
    host = batch[torch.from_numpy(pick).cuda()].cpu().numpy()
    got_h = got[torch.from_numpy(pick).cuda()].cpu().numpy()
    ref_ex = ostain.VahadaneExtractor(random_state=0)
    worst = 0.0
    for j in range(len(pick)):
        exp = ref_ex.get_stain_matrix(host[j].copy())
        worst = max(worst, float(np.abs(got_h[j] - exp).max()))
    assert worst <= 1e-6, worst
    # repeated patches give bit-identical matrices wherever they sit in the launch sequence (both scratch chunks)
    rest = np.setdiff1d(np.arange(n), pick)
    a, b = rest[rest < 256], rest[rest >= n - 256]
    common = np.intersect1d(a % 256, b % 256)[:8]
    for r in common:
        assert torch.equal(got[a[a % 256 == r][0]], got[b[b % 256 == r][0]])
    # StainAugmentor over the whole batch with injected alpha / beta (the reference draws them unseeded, stainaugment.py:232)
    aug = StainAugmentor(method="vahadane", sigma1=0.4, sigma2=0.2, augment_background=False)
    aug.fit(batch, threshold=0.85)
    rng = np.random.default_rng(5)
    alpha = rng.uniform(0.6, 1.4, (n, 2))
    beta = rng.uniform(-0.2, 0.2, (n, 2))
    out = aug.augment(alpha_beta=np.concatenate([alpha, beta], axis=1))
    out_h = out[torch.from_numpy(pick).cuda()].cpu().numpy()
    for j, i in enumerate(pick):
        exp = ostain.stain_augment(host[j], ref_ex.get_stain_matrix(host[j].copy()), alpha[i], beta[i], threshold=0.85,
                                   augment_background=False)
        diff = np.abs(out_h[j].astype(int) - exp.astype(int))
        assert diff.max() <= 1 and (diff != 0).mean() < 2e-3, (i, diff.max(), (diff != 0).mean())
