"""The hand-written ResNet stem (``tia_stem_conv7x7_pool_nhwc``) against the unfused torch ops on the CPU in float32.

Reference path: ``ToTensor`` (uint8 -> float32 / 255, ``models/dataset/classification.py:27-32``) -> ``conv1`` -> ``bn1`` ->
``relu`` -> ``maxpool`` of torchvision's ResNet behind ``CNNModel.forward`` (``models/architecture/vanilla.py:300-316``).
"""

from __future__ import annotations

import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F  # noqa: N812


def _stem_parts(seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    conv = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=True)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.05)
        conv.bias.copy_(torch.randn(64, generator=g) * 0.1)
    return conv


def _reference(conv, x_float_nhwc: torch.Tensor) -> torch.Tensor:
    with torch.inference_mode():
        y = F.conv2d(x_float_nhwc.permute(0, 3, 1, 2), conv.weight, conv.bias, 2, 3)
        return F.max_pool2d(F.relu(y), 3, 2, 1)


# (n, h, w): the benchmark sizes, odd sizes (partial tiles, odd conv / pooled extents), a wide image (three column strips),
# a tall single image (row chunks with their warm-up iteration), tiny images
SHAPES = [(3, 224, 224), (2, 256, 256), (2, 37, 53), (1, 70, 600), (1, 520, 64), (5, 8, 8), (2, 7, 9), (1, 131, 258)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES)
def test_stem_kernel_matches_unfused_torch_ops(shape):
    from tiatoolbox_amd.models.architecture.fused import hip_stem_conv_pool, pack_stem_weights

    n, h, w = shape
    conv = _stem_parts(seed=h * 1000 + w)
    g = torch.Generator().manual_seed(n + h + w)
    x = torch.randint(0, 256, (n, h, w, 3), generator=g, dtype=torch.uint8)
    ref = _reference(conv, x.float().div(255))
    conv_d = copy.deepcopy(conv).cuda()
    wp = pack_stem_weights(conv_d)
    assert wp.shape == (148, 64) and float(wp[147].abs().max()) == 0.0
    got = hip_stem_conv_pool(x.cuda(), wp, conv_d.bias.detach())
    assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    err = (got.cpu() - ref).abs().max().item()
    assert err <= 1e-5, err  # float32 summation order only
    # the pre-pool activation as a second output (the UNet's first skip connection); the pooled output does not change
    got2, conv_out = hip_stem_conv_pool(x.cuda(), wp, conv_d.bias.detach(), return_conv=True)
    with torch.inference_mode():
        ref_conv = F.relu(F.conv2d(x.float().div(255).permute(0, 3, 1, 2), conv.weight, conv.bias, 2, 3))
    assert torch.equal(got2, got) and conv_out.shape == ref_conv.shape
    assert (conv_out.cpu() - ref_conv).abs().max().item() <= 1e-5
    assert torch.equal(F.max_pool2d(conv_out, 3, 2, 1), got)  # the two outputs are the same numbers
    # float32 input: same values, no 1/255 on load
    xf = x.float().div(255)
    got_f = hip_stem_conv_pool(xf.cuda(), wp, conv_d.bias.detach())
    assert torch.equal(got_f, got)  # the uint8 path divides exactly like torch does
    assert torch.equal(hip_stem_conv_pool(xf.cuda(), wp, conv_d.bias.detach(), return_conv=True)[1], conv_out)
    # unscaled floats (a hook that feeds 0..255) go through as they are
    got_raw = hip_stem_conv_pool(x.float().cuda(), wp, conv_d.bias.detach())
    ref_raw = _reference(conv, x.float())
    assert (got_raw.cpu() - ref_raw).abs().max().item() <= 1e-5 * 255


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float16", "bfloat16"])
@pytest.mark.parametrize("shape", [(3, 224, 224), (2, 256, 256), (2, 37, 53), (1, 70, 600), (1, 520, 64), (5, 8, 8), (1, 131, 258)])
def test_half_stem_kernel_matches_float32_convolution_of_the_rounded_operands(shape, dtype):
    """``tia_stem_conv7x7_pool_nhwc_h`` (half matrix cores, float32 accumulate) against the unfused float32 ops on the CPU applied
    to the SAME half-rounded inputs (``x / 255`` rounded to half) and weights: what remains is the float32 summation order and the
    one rounding of the pooled result."""
    from tiatoolbox_amd.models.architecture.fused import hip_stem_conv_pool_h, pack_stem_weights_h

    dt = getattr(torch, dtype)
    eps = 2.0 ** -10 if dt == torch.float16 else 2.0 ** -7
    n, h, w = shape
    conv = _stem_parts(seed=h * 1000 + w + 1)
    g = torch.Generator().manual_seed(n + h + w)
    x = torch.randint(0, 256, (n, h, w, 3), generator=g, dtype=torch.uint8)
    xh = x.float().div(255).to(dt).float()
    wh = conv.weight.detach().to(dt).float()
    with torch.inference_mode():
        ref = F.max_pool2d(F.relu(F.conv2d(xh.permute(0, 3, 1, 2), wh, conv.bias, 2, 3)), 3, 2, 1)
    wp = pack_stem_weights_h(conv.weight.detach().cuda(), dt)
    assert wp.shape == (22, 64, 8) and wp.dtype == dt and float(wp[21].float().abs().max()) == 0.0  # k >= 168: zero padding
    bias = conv.bias.detach().cuda()
    got = hip_stem_conv_pool_h(x.cuda(), wp, bias, dtype=dt)
    assert got.dtype == dt and got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    err = (got.float().cpu() - ref).abs()
    assert bool((err <= eps * ref.abs() + 1e-4).all()), float((err - eps * ref.abs()).max())
    # float32 input holding the same values: rounded to half on staging, same result
    got_f = hip_stem_conv_pool_h(x.float().div(255).cuda(), wp, bias, dtype=dt)
    assert torch.equal(got_f, got)
    # sliced batch (base address not 4-byte aligned)
    if n > 1:
        got_v = hip_stem_conv_pool_h(x.cuda()[1:], wp, bias, dtype=dt)
        assert torch.equal(got_v, got[1:])


@pytest.mark.gpu
def test_stem_kernel_unaligned_view_and_all_byte_values():
    """A batch view whose base address is not 4-byte aligned (odd image size, sliced batch), and every byte value."""
    from tiatoolbox_amd.models.architecture.fused import hip_stem_conv_pool, pack_stem_weights

    conv = _stem_parts(seed=3)
    g = torch.Generator().manual_seed(11)
    x = torch.randint(0, 256, (4, 37, 53, 3), generator=g, dtype=torch.uint8)
    x[0].view(-1)[:256] = torch.arange(256, dtype=torch.uint8)
    conv_d = copy.deepcopy(conv).cuda()
    wp = pack_stem_weights(conv_d)
    xd = x.cuda()
    for first in (1, 2, 3):
        view = xd[first:]
        assert view.data_ptr() % 4 == (first * 37 * 53 * 3) % 4
        got = hip_stem_conv_pool(view, wp, conv_d.bias.detach())
        ref = _reference(conv, x[first:].float().div(255))
        assert (got.cpu() - ref).abs().max().item() <= 1e-5
    # the kernel's 1/255 table == torch's division for every byte
    ramp = torch.arange(256, dtype=torch.uint8).repeat(3, 1).t().reshape(1, 16, 16, 3).contiguous()
    ident = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=True)
    with torch.no_grad():
        ident.weight.zero_()
        ident.bias.zero_()
        ident.weight[0, 0, 3, 3] = 1.0  # output channel 0 = red value of the centre tap
    ident_d = copy.deepcopy(ident).cuda()
    wpi = pack_stem_weights(ident_d)
    got = hip_stem_conv_pool(ramp.cuda(), wpi, ident_d.bias.detach())
    ref = _reference(ident, ramp.float().div(255))
    assert torch.equal(got.cpu(), ref)


@pytest.mark.gpu
def test_stem_wrapper_refuses_what_it_cannot_take():
    from tiatoolbox_amd import _lib
    from tiatoolbox_amd.models.architecture.fused import hip_stem_conv_pool, pack_stem_weights

    conv = _stem_parts().cuda()
    wp = pack_stem_weights(conv)
    with pytest.raises(ValueError, match="NHWC"):
        hip_stem_conv_pool(torch.zeros((1, 8, 8, 3), dtype=torch.uint8), wp, conv.bias.detach())  # host tensor
    with pytest.raises(ValueError, match="NHWC"):
        hip_stem_conv_pool(torch.zeros((1, 3, 8, 8), dtype=torch.uint8, device="cuda"), wp, conv.bias.detach())
    with pytest.raises(ValueError, match="stem kernel"):
        pack_stem_weights(torch.nn.Conv2d(3, 64, 3, 1, 1).cuda())
    assert _lib.load().tia_stem_conv7x7_pool_nhwc(0, 1, 0, 0, 0, 0, 0, 1, 8, 8, None) != 0


@pytest.mark.gpu
def test_float32_classifier_run_launches_only_handwritten_convolutions():
    """A float32 GPU run of the patch classifier contains no library convolution / tensor-op kernel: stem and block
    convolutions are ``stem7x7_pool_kernel`` / ``conv_mfma_f32_kernel``; what is left to torch is the average pool, the
    classifier GEMM and softmax."""
    from torch.profiler import ProfilerActivity, profile

    from tiatoolbox_amd.models.engine.patch_predictor import PatchPredictor
    from tiatoolbox_amd.tools.stainnorm import get_normalizer
    from tiatoolbox_amd.utils import synth

    patches = torch.from_numpy(synth.g_he(8, 224, 224, seed=5)).cuda()
    norm = get_normalizer("macenko")
    norm.fit(synth.g_he(1, 256, 256, seed=6)[0])
    eng = PatchPredictor("resnet18-kather100k", batch_size=8, device="cuda", verbose=False)
    eng.run(patches, patch_mode=True, return_probabilities=True, stain_normalizer=norm)  # builds the inference copy
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        out = eng.run(patches, patch_mode=True, return_probabilities=True, stain_normalizer=norm)
        torch.cuda.synchronize()
    assert np.isfinite(out["probabilities"]).all()
    names = {e.name for e in prof.events() if e.device_type is not None and "cuda" in str(e.device_type).lower()}
    kernels = {n for n in names if "memcpy" not in n.lower() and "memset" not in n.lower()}
    assert any("stem7x7_pool_kernel" in k for k in kernels), kernels
    assert any("conv_mfma_f32_kernel" in k for k in kernels), kernels
    banned = ("igemm", "naive_conv", "SubTensorOp", "ck::", "miopen", "MIOpen", "Im2Col", "gemm_conv", "winograd", "Winograd")
    offenders = {k for k in kernels if any(b in k for b in banned)}
    assert not offenders, offenders
