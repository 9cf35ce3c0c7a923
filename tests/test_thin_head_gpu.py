"""The thin-input form of the MFMA convolution (HoVer-Net's RGB 7x7 stem, reference ``models/architecture/hovernet.py:287-300``)
and the class-head kernel (``hovernet.py:196-199`` ``u0``: BN -> ReLU -> 1x1; ``unet.py:336`` ``clf``) against the unfused
torch ops on the CPU in float32; and: a float32 GPU run of the two segmentation networks launches no library convolution."""

from __future__ import annotations

import numpy as np
import pytest
import torch
import torch.nn.functional as F  # noqa: N812


@pytest.mark.gpu
@pytest.mark.parametrize(("shape", "kernel", "stride", "pads", "cout"), [
    ((2, 3, 64, 64), 7, 1, (3, 3), 64),      # HoVer-Net fast: "same"
    ((1, 3, 70, 53), 7, 1, (0, 0), 64),      # HoVer-Net original: valid; odd width
    ((2, 3, 33, 41), 7, 2, (3, 3), 128),     # strided
    ((1, 4, 20, 24), 5, 1, (2, 2), 64),      # 4 channels x 5 taps = 20 floats per row
    ((3, 1, 17, 19), 3, 1, (1, 1), 64),      # single channel
    ((1, 3, 256, 256), 7, 1, (3, 3), 64),    # the tile size of the engine
])
def test_thin_input_convolution_matches_torch(shape, kernel, stride, pads, cout):
    from tiatoolbox_amd.models.architecture.fused import hip_conv2d_thin, pack_thin_conv_weights

    g = torch.Generator().manual_seed(sum(shape) + kernel)
    n, c, h, w = shape
    x = torch.rand(shape, generator=g)
    wgt = torch.randn((cout, c, kernel, kernel), generator=g) * 0.1
    bias = torch.randn(cout, generator=g) * 0.1
    ref = F.relu(F.conv2d(F.pad(x, (pads[0], pads[1], pads[0], pads[1])), wgt, bias, stride))
    packed = pack_thin_conv_weights(wgt.cuda())
    assert packed.shape == (kernel, 32, cout) and float(packed[:, kernel * c:].abs().max()) == 0.0
    got = hip_conv2d_thin(x.cuda().contiguous(memory_format=torch.channels_last), packed, bias.cuda(), kernel=kernel, stride=stride,
                          pad_lo=pads[0], pad_hi=pads[1], relu=True)
    assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert (got.cpu() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    # a plain NCHW-contiguous input is accepted as well (the wrapper re-lays the rows anyway)
    got2 = hip_conv2d_thin(x.cuda(), packed, bias.cuda(), kernel=kernel, stride=stride, pad_lo=pads[0], pad_hi=pads[1], relu=True)
    assert torch.equal(got2, got)


@pytest.mark.gpu
@pytest.mark.parametrize("cout", [1, 2, 5, 6, 8])
@pytest.mark.parametrize("with_pre", [False, True])
def test_class_head_kernel_matches_torch(cout, with_pre):
    from tiatoolbox_amd.models.architecture.fused import hip_conv1x1_head

    g = torch.Generator().manual_seed(10 * cout + int(with_pre))
    x = torch.randn((3, 64, 37, 29), generator=g)  # 3219 pixels: not a multiple of the 4-pixel wave group
    wgt = torch.randn((cout, 64, 1, 1), generator=g) * 0.2
    bias = torch.randn(cout, generator=g)
    scale, shift = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    xin = F.relu(x * scale[None, :, None, None] + shift[None, :, None, None]) if with_pre else x
    ref = F.conv2d(xin, wgt, bias)
    got = hip_conv1x1_head(x.cuda().contiguous(memory_format=torch.channels_last), wgt.cuda(), bias.cuda(),
                           pre_scale=scale.cuda() if with_pre else None, pre_shift=shift.cuda() if with_pre else None)
    assert got.shape == ref.shape
    assert (got.cpu() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    with pytest.raises(ValueError, match="64 channels"):
        hip_conv1x1_head(torch.zeros((1, 32, 4, 4), device="cuda").contiguous(memory_format=torch.channels_last), wgt.cuda()[:, :32], None)


def _device_kernels(fn) -> set:
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        fn()
        torch.cuda.synchronize()
    names = {e.name for e in prof.events() if e.device_type is not None and "cuda" in str(e.device_type).lower()}
    return {n for n in names if "memcpy" not in n.lower() and "memset" not in n.lower()}


BANNED = ("igemm", "naive_conv", "SubTensorOp", "ck::", "miopen", "MIOpen", "Im2Col", "gemm_conv", "winograd", "Winograd", "Cijk_")


@pytest.mark.gpu
def test_float32_segmentation_forwards_launch_only_handwritten_convolutions():
    """``FusedHoVerNet`` and ``FusedUNet`` (what the engines run for float32 on the GPU): stem, trunk, decoders and heads
    are ``conv_mfma_f32_kernel`` / ``stem7x7_pool_kernel`` / ``grouped_conv_valid_kernel`` / ``head1x1_kernel`` -- no MIOpen,
    rocBLAS or hipBLASLt kernel in the forward."""
    from tiatoolbox_amd.models.architecture import get_pretrained_model
    from tiatoolbox_amd.models.architecture.hovernet_fused import FusedHoVerNet
    from tiatoolbox_amd.models.architecture.unet_fused import FusedUNet
    from tiatoolbox_amd.utils import synth

    tiles = torch.from_numpy(synth.g_he(2, 256, 256, seed=5)).cuda()
    with torch.inference_mode():
        model, _ = get_pretrained_model("hovernet_fast-pannuke")
        hov = FusedHoVerNet(model.eval().cuda()).cuda()
        xin = tiles.float().permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
        hov(xin)
        kernels = _device_kernels(lambda: hov(xin))
        assert any("conv_mfma_f32_kernel" in k for k in kernels) and any("head1x1_kernel" in k for k in kernels), kernels
        assert any("grouped_conv_valid_kernel" in k for k in kernels), kernels
        assert not {k for k in kernels if any(b in k for b in BANNED)}, kernels
        model, _ = get_pretrained_model("fcn_resnet50_unet-bcss")
        unet = FusedUNet(model.eval().cuda()).cuda()
        xu = tiles.permute(0, 3, 1, 2)
        out = unet(xu)
        assert np.isfinite(out.cpu().numpy()).all()
        kernels = _device_kernels(lambda: unet(xu))
        assert any("stem7x7_pool_kernel" in k for k in kernels) and any("head1x1_kernel" in k for k in kernels), kernels
        assert not {k for k in kernels if any(b in k for b in BANNED)}, kernels


@pytest.mark.gpu
def test_segmentation_forwards_with_winograd_match_direct():
    """``conv_algo="winograd"`` on the fused segmentation networks: every plain 3x3 / stride-1 MFMA convolution of ``FusedHoVerNet``
    (the residual units' conv2) and ``FusedUNet`` (Bottleneck conv2, the decoder's 3x3 layers; "same" and valid borders) goes through
    ``conv3x3_wino_kernel``; head maps / logits stay within 1e-4 of the direct float32 forward (relative to the largest magnitude),
    and the switch goes back."""
    from tiatoolbox_amd.models.architecture import get_pretrained_model
    from tiatoolbox_amd.models.architecture.hovernet_fused import FusedHoVerNet, set_conv_algo
    from tiatoolbox_amd.models.architecture.unet_fused import FusedUNet
    from tiatoolbox_amd.utils import synth

    tiles = torch.from_numpy(synth.g_he(2, 256, 256, seed=7)).cuda()
    with torch.inference_mode():
        model, _ = get_pretrained_model("hovernet_fast-pannuke")
        hov = FusedHoVerNet(model.eval().cuda()).cuda()
        xin = tiles.float().permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
        direct = {k: v.clone() for k, v in hov(xin).items()}
        assert set_conv_algo(hov, "winograd") >= 10  # noqa: PLR2004
        kernels = _device_kernels(lambda: hov(xin))
        assert any("conv3x3_wino_kernel" in k for k in kernels), kernels
        wino = hov(xin)
        for k, v in direct.items():
            scale = v.abs().max().item()
            assert (wino[k] - v).abs().max().item() <= 1e-4 * max(scale, 1.0), k
        assert any(not torch.equal(wino[k], direct[k]) for k in direct)
        set_conv_algo(hov, "direct")
        again = hov(xin)
        assert all(torch.equal(again[k], direct[k]) for k in direct)
        with pytest.raises(ValueError, match="conv_algo"):
            set_conv_algo(hov, "fft")
        model, _ = get_pretrained_model("fcn_resnet50_unet-bcss")
        unet = FusedUNet(model.eval().cuda()).cuda()
        xu = tiles.permute(0, 3, 1, 2)
        ref = unet(xu).clone()
        assert set_conv_algo(unet, "winograd") >= 16  # noqa: PLR2004
        got = unet(xu)
        assert (got - ref).abs().max().item() <= 1e-4 * max(ref.abs().max().item(), 1.0)
        assert not torch.equal(got, ref)
