"""Host-side check of the fused inference graphs (``hovernet_fused.py``, ``unet_fused.py``) WITHOUT a GPU.

The HIP entry points they call are replaced by their plain-torch definitions (what each kernel computes, stated in
``include/tiatoolbox_amd.h``), so what is tested here is the graph surgery itself: BN folding, which ReLU / residual goes
into which epilogue, the TF-"same" padding arithmetic, the in-place feature stack of the dense units, the shared
up-sampling.  The kernels themselves are compared with torch in the ``-m gpu`` tests
(``test_fused_hovernet_forward_matches_plain_module``, ``test_fused_unet_forward_matches_plain_module``).
"""

from __future__ import annotations

import copy

import pytest
import torch
import torch.nn.functional as F  # noqa: N812


def _conv_ex(x, wp, bias, res, *, kernel, stride, pad_lo, pad_hi, relu):  # noqa: ARG001
    y = F.conv2d(F.pad(x, (pad_lo, pad_hi, pad_lo, pad_hi)), wp.permute(3, 2, 0, 1), bias, stride)
    y = y + res if res is not None else y
    return F.relu(y) if relu else y


def _conv_post(x, wp, bias, res, *, kernel, stride, pad_lo, pad_hi, relu, post_scale, post_shift, want_raw=True):
    v = _conv_ex(x, wp, bias, res, kernel=kernel, stride=stride, pad_lo=pad_lo, pad_hi=pad_hi, relu=relu)
    return (v if want_raw else None), F.relu(v * post_scale[None, :, None, None] + post_shift[None, :, None, None])


def _conv_pre(x, pre_scale, pre_shift, wp, bias, res=None, *, stride=1, relu=False):
    a = F.relu(x * pre_scale[None, :, None, None] + pre_shift[None, :, None, None])
    return _conv_ex(a, wp, bias, res, kernel=1, stride=stride, pad_lo=0, pad_hi=0, relu=relu)


def _scale_shift(x, sc, sh, *, relu=True, inplace=False):  # noqa: ARG001
    y = x * sc[None, :, None, None] + sh[None, :, None, None]
    return F.relu(y) if relu else y


def _bias_act(y, bias, res=None, *, relu=True):
    y = y + bias[None, :, None, None]
    y = y + res if res is not None else y
    return F.relu(y) if relu else y


def _grouped(x, wp, *, groups, kernel, out=None):
    y = F.conv2d(x, wp.permute(0, 4, 3, 1, 2).reshape(groups * 8, 32, kernel, kernel), None, 1, 0, 1, groups)
    if out is not None:
        out.copy_(y)
        return out
    return y


@pytest.fixture
def torch_kernels(monkeypatch):
    import tiatoolbox_amd.models.architecture.hovernet_fused as hf
    import tiatoolbox_amd.models.architecture.unet_fused as uf

    monkeypatch.setattr(hf, "hip_conv2d_ex", _conv_ex)
    monkeypatch.setattr(hf, "hip_conv2d_post", _conv_post)
    monkeypatch.setattr(hf, "hip_conv1x1_pre", _conv_pre)
    monkeypatch.setattr(hf, "pack_conv_weights", lambda conv: conv.weight.detach().permute(2, 3, 1, 0).contiguous())
    monkeypatch.setattr(hf, "hip_scale_shift_act", _scale_shift)
    monkeypatch.setattr(hf, "hip_scale_shift_act_view", lambda x, sc, sh, relu=True: _scale_shift(x, sc, sh, relu=relu))
    monkeypatch.setattr(hf, "hip_bias_act_", _bias_act)
    monkeypatch.setattr(hf, "hip_grouped_conv_valid", _grouped)
    def up(x, y, scale=None, shift=None):
        out = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3) + y
        return out if scale is None else F.relu(out * scale[None, :, None, None] + shift[None, :, None, None])

    def stem(x_nhwc, wp, bias, *, out_dtype=torch.float32, return_conv=False):  # noqa: ARG001
        w = wp[:147].view(7, 7, 3, 64).permute(3, 2, 0, 1)
        xf = x_nhwc.float().div(255) if x_nhwc.dtype == torch.uint8 else x_nhwc
        conv = F.relu(F.conv2d(xf.permute(0, 3, 1, 2), w, bias, 2, 3))
        pooled = F.max_pool2d(conv, 3, 2, 1)
        return (pooled, conv) if return_conv else pooled

    def pack_stem(weight):
        return torch.cat([weight.detach().permute(2, 3, 1, 0).reshape(147, 64), torch.zeros(1, 64)])

    def thin(x, wp, bias, *, kernel, stride, pad_lo, pad_hi, relu):
        c = x.shape[1]
        w = wp[:, :kernel * c].reshape(kernel, kernel, c, -1).permute(3, 2, 0, 1)
        y = F.conv2d(F.pad(x, (pad_lo, pad_hi, pad_lo, pad_hi)), w, bias, stride)
        return F.relu(y) if relu else y

    def pack_thin(weight):
        cout, c, kh, kw = weight.shape
        packed = torch.zeros((kh, 32, cout))
        packed[:, :kw * c] = weight.detach().permute(2, 3, 1, 0).reshape(kh, kw * c, cout)
        return packed

    def head(x, weight, bias, *, pre_scale=None, pre_shift=None):
        if pre_scale is not None:
            x = F.relu(x * pre_scale[None, :, None, None] + pre_shift[None, :, None, None])
        return F.conv2d(x, weight.reshape(weight.shape[0], 64, 1, 1), bias)

    monkeypatch.setattr(hf, "hip_conv2d_thin", thin)
    monkeypatch.setattr(hf, "pack_thin_conv_weights", pack_thin)
    monkeypatch.setattr(hf, "hip_conv1x1_head", head)
    monkeypatch.setattr(hf, "hip_upsample2x_add", up)
    monkeypatch.setattr(uf, "hip_upsample2x_add", up)
    monkeypatch.setattr(uf, "hip_stem_conv_pool", stem)
    monkeypatch.setattr(uf, "pack_stem_weights", pack_stem)
    return hf, uf


def _randomise_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.05, generator=g)
            mod.running_var.uniform_(0.8, 1.2, generator=g)
            mod.weight.data.uniform_(0.8, 1.2, generator=g)
            mod.bias.data.normal_(0, 0.05, generator=g)
    return g


@pytest.mark.parametrize(("mode", "size"), [("fast", 256), ("original", 270)])
def test_fused_hovernet_graph_equals_forward(torch_kernels, mode, size):
    from tiatoolbox_amd.models.architecture.hovernet import HoVerNet

    hf, _ = torch_kernels
    torch.manual_seed(3)
    model = HoVerNet(num_types=6, mode=mode).eval()
    g = _randomise_bn(model, 5)
    x = torch.randint(0, 256, (1, 3, size, size), generator=g).float()
    with torch.inference_mode():
        ref = model(x)
        fused = hf.FusedHoVerNet(copy.deepcopy(model))
        got = fused(x)
    assert sum(1 for m in fused.modules() if getattr(m, "mfma_ok", False)) == 104
    assert sum(1 for m in fused.modules() if getattr(m, "grouped_ok", False)) == 36
    assert list(got) == list(ref) == ["tp", "np", "hv"]
    for name in ref:
        assert got[name].shape == ref[name].shape
        assert float((got[name] - ref[name]).abs().max()) <= 2e-6, name


def test_fused_hovernetplus_and_unet_graphs_equal_forward(torch_kernels):
    from tiatoolbox_amd.models.architecture.hovernetplus import HoVerNetPlus
    from tiatoolbox_amd.models.architecture.unet import UNetModel

    hf, uf = torch_kernels
    torch.manual_seed(2)
    plus = HoVerNetPlus(num_types=3, num_layers=5).eval()
    g = _randomise_bn(plus, 7)
    x = torch.randint(0, 256, (1, 3, 256, 256), generator=g).float()
    with torch.inference_mode():
        ref, got = plus(x), hf.FusedHoVerNet(copy.deepcopy(plus))(x)
    assert list(got) == list(ref) == ["tp", "np", "hv", "ls"]
    assert max(float((got[k] - ref[k]).abs().max()) for k in ref) <= 2e-6

    unet = UNetModel(3, 5, "resnet50", decoder_block=[3, 3]).eval()
    g = _randomise_bn(unet, 9)
    x = torch.randint(0, 256, (1, 3, 192, 256), generator=g).float()
    with torch.inference_mode():
        ref = unet(x)
        fused = uf.FusedUNet(copy.deepcopy(unet))
        got = fused(x)
    assert sum(1 for m in fused.modules() if getattr(m, "mfma_ok", False)) == 61
    assert got.shape == ref.shape == (1, 5, 96, 128) and float((got - ref).abs().max()) <= 2e-6
    with pytest.raises(TypeError, match="ResNet-50 encoder"):
        uf.FusedUNet(UNetModel(3, 2, "unet"))


def test_same_padding_arithmetic_matches_the_layer():
    """``_same_pads`` == ``TFSamepaddingLayer`` (reference ``hovernet.py:30-69``) for the sizes / strides the network uses."""
    from tiatoolbox_amd.models.architecture.hovernet import TFSamepaddingLayer
    from tiatoolbox_amd.models.architecture.hovernet_fused import _same_pads

    for size in (7, 8, 63, 64, 255, 256):
        for k, s in ((3, 1), (3, 2), (5, 1), (7, 1), (7, 2)):
            lo, hi = _same_pads(size, k, s)
            padded = TFSamepaddingLayer(k, s)(torch.zeros(1, 1, size, size))
            assert padded.shape[2] == size + lo + hi
            probe = TFSamepaddingLayer(k, s)(torch.ones(1, 1, size, size))[0, 0]
            assert probe[:lo].sum() == 0 and probe[lo].sum() > 0  # exactly `lo` zero rows in front


def test_hip_wrappers_refuse_host_tensors():
    """The convolution / epilogue wrappers are device-only: a CPU tensor is an error, never a silent torch fallback."""
    from tiatoolbox_amd.models.architecture import fused

    x = torch.zeros((1, 32, 4, 4)).contiguous(memory_format=torch.channels_last)
    w = torch.zeros((1, 1, 32, 64))
    with pytest.raises(ValueError, match="channels-last CUDA"):
        fused.hip_conv2d_ex(x, w, None, None, kernel=1, stride=1, pad_lo=0, pad_hi=0, relu=False)
    with pytest.raises(ValueError, match="channels-last CUDA"):
        fused.hip_conv2d_post(x, w, None, None, kernel=1, stride=1, pad_lo=0, pad_hi=0, relu=False,
                              post_scale=torch.ones(64), post_shift=torch.zeros(64))
    with pytest.raises(ValueError, match="channels-last CUDA"):
        fused.hip_conv1x1_pre(x, torch.ones(32), torch.zeros(32), w, None)
    with pytest.raises(ValueError, match="channels-last CUDA"):
        fused.hip_scale_shift_act(x, torch.ones(32), torch.zeros(32))
    with pytest.raises(ValueError, match="CUDA"):
        fused.hip_upsample2x_add(x, torch.zeros((1, 32, 8, 8)))
    with pytest.raises(ValueError, match="CUDA"):
        fused.hip_scale_shift_act_view(x, torch.ones(32), torch.zeros(32))
    with pytest.raises(ValueError, match="channels-last CUDA"):
        fused.hip_grouped_conv_valid(torch.zeros((1, 128, 5, 5)).contiguous(memory_format=torch.channels_last),
                                     torch.zeros((4, 3, 3, 32, 8)), groups=4, kernel=3)
