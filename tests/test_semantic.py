"""Semantic segmentation path: patch grids, canvas stitching, UNet, SemanticSegmentor."""

from __future__ import annotations

from pathlib import Path

import numpy as np
import pytest

from oracle import semantic as osem
from tiatoolbox_amd.tools.patchextraction import PatchExtractor

GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD / "grid_golden.npz")


class _Mask:
    def __init__(self, img):
        self.img = img


def test_patch_grid_matches_real_reference(gold):
    for tag in "abc":
        img, pin, pout, stride = (tuple(int(v) for v in row) for row in gold[f"{tag}_args"])
        i_b, o_b = PatchExtractor.get_coordinates(patch_output_shape=pout, image_shape=img, patch_input_shape=pin,
                                                  stride_shape=stride)
        assert np.array_equal(i_b, gold[f"{tag}_in"]) and np.array_equal(o_b, gold[f"{tag}_out"])
        for ratio in (0.0, 0.4):
            keep = PatchExtractor.filter_coordinates(_Mask(gold[f"{tag}_mask"]), o_b, img, min_mask_ratio=ratio)
            assert np.array_equal(keep, gold[f"{tag}_keep{int(ratio * 10)}"]), (tag, ratio)
    with pytest.raises(ValueError, match="must larger than"):
        PatchExtractor.get_coordinates(patch_output_shape=(9, 9), image_shape=(50, 50), patch_input_shape=(8, 8),
                                       stride_shape=(4, 4))
    with pytest.raises(ValueError, match="`coordinates_list` should be ndarray of integer type"):
        PatchExtractor.filter_coordinates(_Mask(np.ones((4, 4))), np.zeros((2, 4)), (10, 10))
    with pytest.raises(ValueError, match="min_mask_ratio"):
        PatchExtractor.filter_coordinates(_Mask(np.ones((4, 4))), np.zeros((2, 4), int), (10, 10), min_mask_ratio=2)


def _known_answers(merge):
    """Reference tests/engines/test_semantic_segmentor.py:283-333."""
    canvas, count = merge(np.array([np.ones((2, 2, 1)), np.ones((2, 2, 1)) * 2]), np.array([[0, 0, 2, 2], [2, 0, 4, 2]]), (2, 4, 1))
    assert np.array_equal(canvas[:, :2, :], np.ones((2, 2, 1))) and np.array_equal(canvas[:, 2:, :], np.ones((2, 2, 1)) * 2)
    assert np.array_equal(count, np.ones((2, 4, 1)))
    canvas, count = merge(np.array([np.ones((2, 2, 1)), np.ones((2, 2, 1)) * 3]), np.array([[0, 0, 2, 2], [1, 0, 3, 2]]), (2, 3, 1))
    assert np.array_equal(canvas, np.array([[[1], [4], [3]], [[1], [4], [3]]]))
    assert np.array_equal(count, np.array([[[1], [2], [1]], [[1], [2], [1]]]))
    canvas, count = merge(np.array([np.zeros((2, 2, 1)), np.ones((2, 2, 1))]), np.array([[0, 0, 2, 2], [2, 0, 4, 2]]), (2, 4, 1))
    assert np.array_equal(canvas[:, :2, :], np.zeros((2, 2, 1))) and np.array_equal(canvas[:, 2:, :], np.ones((2, 2, 1)))
    assert np.array_equal(count[:, :2, :], np.zeros((2, 2, 1))) and np.array_equal(count[:, 2:, :], np.ones((2, 2, 1)))
    canvas, count = merge(np.empty((0, 2, 2, 1)), np.empty((0, 4)), (2, 2, 1))
    assert np.array_equal(canvas, np.zeros((2, 2, 1))) and np.array_equal(count, np.zeros((2, 2, 1), dtype=np.uint8))
    assert count.dtype == np.uint8


def test_oracle_merge_known_answers_and_real_reference(gold):
    _known_answers(osem.merge_batch_to_canvas)
    canvas, count = osem.merge_batch_to_canvas(gold["merge_blocks"], gold["merge_locs"], (16, 60, 3))
    assert np.array_equal(canvas, gold["merge_canvas"]) and np.array_equal(count, gold["merge_count"])


def test_unet_state_dict_and_shapes():
    import torch

    from tiatoolbox_amd.models.architecture.unet import UNetModel

    m = UNetModel(3, 5, "resnet50", decoder_block=[3, 3]).eval()
    keys = set(m.state_dict())
    for k in ("backbone.conv1.weight", "backbone.layer4.2.bn3.running_var", "backbone.fc.bias", "conv1x1.weight",
              "uplist.0.0.weight", "uplist.3.5.weight", "clf.bias", "upsample2x.unpool_mat"):
        assert k in keys, k
    out = m.infer_batch(m, torch.rand(1, 128, 128, 3) * 255, device="cpu")
    assert out.shape == (1, 64, 64, 5)
    np.testing.assert_allclose(out.sum(-1), 1.0, atol=1e-5)
    small = UNetModel(3, 2, "unet", encoder_levels=[4, 8, 16], skip_type="concat").eval()
    assert small.infer_batch(small, torch.rand(1, 32, 32, 3) * 255, device="cpu").shape == (1, 48, 48, 2)
    with pytest.raises(ValueError, match="Unknown encoder"):
        UNetModel(encoder="vgg")


# ------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_hip_merge_known_answers_and_real_reference(gold):
    from tiatoolbox_amd.models.engine.semantic_segmentor import merge_batch_to_canvas

    _known_answers(merge_batch_to_canvas)
    canvas, count = merge_batch_to_canvas(gold["merge_blocks"], gold["merge_locs"], (16, 60, 3))
    assert np.array_equal(canvas, gold["merge_canvas"]) and np.array_equal(count, gold["merge_count"])


@pytest.mark.gpu
def test_hip_wsi_stitch_bit_exact_vs_oracle():
    """Random blocks on a masked grid: probabilities and argmax equal the NumPy restatement exactly."""
    import torch

    from tiatoolbox_amd.models.engine import semantic_segmentor as ss

    rng = np.random.default_rng(0)
    h, w, oh, stride, c = 300, 410, 64, 50, 5
    _, out_b = PatchExtractor.get_coordinates(patch_output_shape=(oh, oh), image_shape=(w, h),
                                              patch_input_shape=(oh * 2, oh * 2), stride_shape=(stride, stride))
    keep = rng.random(len(out_b)) < 0.8
    blocks = rng.random((len(out_b), oh, oh, c)).astype(np.float32)
    blocks /= blocks.sum(-1, keepdims=True)
    blocks[~keep] = 0
    exp = osem.merge_wsi(blocks, out_b, (h, w))
    dev = torch.device("cuda")
    pred = torch.zeros((h, w), dtype=torch.uint8, device=dev)
    probs = torch.zeros((h, w, c), dtype=torch.float32, device=dev)
    prev = None
    ys_list = np.unique(out_b[:, 1])
    for i, ys in enumerate(ys_list):
        sel = np.flatnonzero(out_b[:, 1] == ys)
        row, cnt = ss._row_merge(torch.from_numpy(blocks[sel]).to(dev), out_b[sel, 0], w)
        y1 = min(int(ys_list[i + 1]) if i + 1 < len(ys_list) else int(ys) + oh, h)
        if prev is None:
            ss._finalize(row, cnt, int(ys), None, None, 0, int(ys), y1, probs, pred)
        else:
            ss._finalize(prev[0], prev[1], prev[2], row, cnt, int(ys), int(ys), y1, probs, pred)
        prev = (row, cnt, int(ys))
    assert np.array_equal(probs.cpu().numpy(), exp)
    assert np.array_equal(pred.cpu().numpy(), exp.argmax(-1).astype(np.uint8))


@pytest.mark.gpu
def test_semantic_segmentor_wsi_mode_matches_oracle_composition(conv_algo):
    """Engine (tiling + tissue mask + UNet + device stitching) == same patches through the oracle merge."""
    import torch

    from tiatoolbox_amd.models.architecture.unet import UNetModel
    from tiatoolbox_amd.models.engine.io_config import IOSegmentorConfig
    from tiatoolbox_amd.models.engine.semantic_segmentor import SemanticSegmentor
    from tiatoolbox_amd.utils import synth
    from tiatoolbox_amd.wsicore import ArrayWSIReader

    torch.manual_seed(0)
    model = UNetModel(3, 3, "resnet50").eval()
    cfg = IOSegmentorConfig(input_resolutions=[{"units": "mpp", "resolution": 0.25}],
                            output_resolutions=[{"units": "mpp", "resolution": 0.25}], patch_input_shape=[128, 128],
                            patch_output_shape=[64, 64], stride_shape=[50, 50],
                            save_resolution={"units": "mpp", "resolution": 0.25})
    slide = np.full((600, 700, 3), 245, np.uint8)
    slide[64:480, 96:600] = synth.g_he(1, 416, 504, seed=3)[0]
    eng = SemanticSegmentor(model, batch_size=8, device="cuda")
    reader = ArrayWSIReader(slide, mpp=0.25, power=40)
    import tempfile
    from pathlib import Path

    with tempfile.TemporaryDirectory() as tmp:
        with pytest.raises(OSError, match="no save directory"):  # engine_abc.py:1866-1871
            eng.run([reader], patch_mode=False, ioconfig=cfg)
        paths = eng.run([reader], patch_mode=False, ioconfig=cfg, return_probabilities=True, save_dir=Path(tmp) / "out",
                        conv_algo=conv_algo)
        assert list(paths) == [0] and paths[0].name == "0.npz"
        with np.load(paths[0]) as res:
            pred, probs = res["predictions"], res["probabilities"]
    assert pred.shape == (600, 700) and pred.dtype == np.uint8
    # recompute from the same patches with the oracle merge
    mask_reader = reader.tissue_mask(resolution=1.25, units="power")
    in_b, out_b, keep = eng.get_coordinates(reader, mask_reader)
    assert 0 < keep.sum() < len(keep)  # the mask really drops background tiles
    dev_model = eng._inference_model(torch.float32)
    blocks = np.zeros((len(out_b), 64, 64, 3), np.float32)
    for i in np.flatnonzero(keep):
        blocks[i] = model.infer_batch(dev_model, reader.read_bounds_batch(in_b[i:i + 1]), device="cuda")[0].cpu().numpy()
    exp = osem.merge_wsi(blocks, out_b, (600, 700))
    np.testing.assert_allclose(probs, exp, atol=2e-6)      # batch-size dependent conv algorithms: not bitwise
    agree = (pred == exp.argmax(-1)).mean()
    assert agree > 0.9999, agree
    # patch mode: dense probabilities + argmax predictions
    patches = synth.g_he(3, 128, 128, seed=4)
    out = eng.run(patches, patch_mode=True, ioconfig=cfg, return_probabilities=True)
    assert out["probabilities"].shape == (3, 64, 64, 3) and out["predictions"].shape == (3, 64, 64)


@pytest.mark.gpu
def test_hip_gather_patches_matches_padded_slicing():
    """``tia_gather_patches_u8`` (ArrayWSIReader.read_bounds_batch) == NumPy slicing of a 255-padded slide: regions
    hanging over every edge and corner, fully outside, byte-unaligned x offsets, 1-channel (mask) slides, and the
    odd-size fallback (``WSIPatchDataset.__getitem__`` pads with 255, dataset_abc.py:430-436)."""
    from tiatoolbox_amd.wsicore import ArrayWSIReader

    rng = np.random.default_rng(4)
    slide = rng.integers(0, 255, (157, 203, 3), dtype=np.uint8)
    reader = ArrayWSIReader(slide)

    def expect(arr, bounds, pad_value=255):
        p = 300
        padded = np.pad(arr, ((p, p), (p, p)) + ((0, 0),) * (arr.ndim - 2), constant_values=pad_value)
        return np.stack([padded[y0 + p:y1 + p, x0 + p:x1 + p] for x0, y0, x1, y1 in bounds])

    for pw, ph in ((64, 48), (32, 32), (20, 7), (203, 157), (256, 256)):
        tl = np.array([[0, 0], [-17, -5], [203 - pw + 9, 3], [5, 157 - ph + 11], [-pw + 1, -ph + 1], [203 - 1, 157 - 1],
                       [-pw - 3, 10], [1, 2], [2, 3], [3, 1], [50, 60], [-1, 155]])
        bounds = np.concatenate([tl, tl + np.array([pw, ph])], axis=1)
        got = reader.read_bounds_batch(bounds).cpu().numpy()
        assert got.shape == (len(bounds), ph, pw, 3)
        assert np.array_equal(got, expect(slide, bounds)), (pw, ph)
    odd = np.array([[3, 4, 3 + 5, 4 + 5], [-2, -2, 3, 3]])  # 5*5*3 bytes: not a dword multiple
    assert np.array_equal(reader.read_bounds_batch(odd).cpu().numpy(), expect(slide, odd))
    assert np.array_equal(reader.read_bounds(odd[0]), expect(slide, odd[:1])[0])
    mask = rng.integers(0, 2, (64, 72), dtype=np.uint8)
    mreader = ArrayWSIReader(mask, mpp=None, power=1.25, mode="bool")
    mb = np.array([[-4, -4, 12, 12], [60, 50, 76, 66]])
    assert np.array_equal(mreader.read_bounds_batch(mb, pad_value=0).cpu().numpy(), expect(mask, mb, 0))
    with pytest.raises(ValueError, match="one size"):
        reader.read_bounds_batch(np.array([[0, 0, 4, 4], [0, 0, 8, 8]]))
    # thumbnail: exact box mean, rounded half to even (cv2.INTER_AREA at an integer factor)
    big = rng.integers(0, 255, (64, 96, 3), dtype=np.uint8)
    thumb = ArrayWSIReader(big, power=40.0).slide_thumbnail(5.0, "power").cpu().numpy()
    exp = np.rint(big.reshape(8, 8, 12, 8, 3).astype(np.float64).mean(axis=(1, 3))).astype(np.uint8)
    assert np.array_equal(thumb, exp)
    # a factor that is not a power of two (40x -> 8x = 5; trailing rows / columns that do not fill a box are dropped): OpenCV's
    # ResizeAreaFast arithmetic = integer box sum * float32(1 / 25), cvRound
    odd = rng.integers(0, 256, (67, 93, 3), dtype=np.uint8)
    thumb5 = ArrayWSIReader(odd, power=40.0).slide_thumbnail(8.0, "power").cpu().numpy()
    sums = odd[:65, :90].reshape(13, 5, 18, 5, 3).astype(np.int64).sum(axis=(1, 3))
    exp5 = np.rint(sums.astype(np.float32) * np.float32(1.0 / 25.0)).astype(np.uint8)
    assert thumb5.shape == (13, 18, 3) and np.array_equal(thumb5, exp5)
    grey = ArrayWSIReader(odd[..., 0].copy(), mpp=None, power=40.0, mode="bool").slide_thumbnail(8.0, "power").cpu().numpy()
    assert np.array_equal(grey, exp5[..., 0])
    with pytest.raises(ValueError, match="integer down-sampling"):
        ArrayWSIReader(big, power=40.0).slide_thumbnail(3.0, "power")


@pytest.mark.gpu
def test_fused_unet_forward_matches_plain_module():
    """``FusedUNet`` (stem on the stem kernel, 61 of the other 62 convolutions on the MFMA kernel: Bottlenecks as conv+BN+ReLU / conv+BN+identity+ReLU
    launches, up-sampling fused with the skip add, decoder pre-activations in one pass) against the plain torch module
    on the CPU in float32 with randomised BN statistics: logits within 2e-4 of their range; and it is what the engine
    runs for float32 on the GPU."""
    import copy

    import torch

    from tiatoolbox_amd.models.architecture.unet import UNetModel
    from tiatoolbox_amd.models.architecture.unet_fused import FusedUNet
    from tiatoolbox_amd.models.engine.semantic_segmentor import SemanticSegmentor

    torch.manual_seed(1)
    g = torch.Generator().manual_seed(5)
    model = UNetModel(3, 5, "resnet50", decoder_block=[3, 3]).eval()
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.05, generator=g)
            mod.running_var.uniform_(0.8, 1.2, generator=g)
            mod.weight.data.uniform_(0.8, 1.2, generator=g)
            mod.bias.data.normal_(0, 0.05, generator=g)
    x = torch.randint(0, 256, (2, 3, 256, 320), generator=g).float()
    with torch.inference_mode():
        ref = model(x)
        got = FusedUNet(copy.deepcopy(model).cuda()).cuda()(x.cuda().contiguous(memory_format=torch.channels_last)).cpu()
    assert got.shape == ref.shape == (2, 5, 128, 160)
    assert (got - ref).abs().max() <= 2e-4 * max(float(ref.abs().max()), 1.0)
    # the uint8 batch as `infer_batch` hands it over (NCHW view of the NHWC bytes): the stem kernel's x / 255 on load
    with torch.inference_mode():
        fused = FusedUNet(copy.deepcopy(model).cuda()).cuda()
        xb = x.to(torch.uint8).permute(0, 2, 3, 1).contiguous().cuda()
        got_u8 = fused(xb.permute(0, 3, 1, 2)).cpu()
    # (not torch.equal: the final 64 -> n_classes 1x1 is a library call whose solver may differ between the two calls)
    assert (got_u8 - got).abs().max() <= 1e-5 * max(float(ref.abs().max()), 1.0)
    eng = SemanticSegmentor(model, batch_size=2, device="cuda")
    assert type(eng._inference_model(torch.float32)).__name__ == "FusedUNet"
