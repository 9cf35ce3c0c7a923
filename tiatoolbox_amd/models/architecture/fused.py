"""Inference-time graph surgery for the CNN backbones (eval mode only).

* ``fold_conv_bn``: BatchNorm folded into the preceding convolution (weights scaled, bias added), removing one full
  read+write of every activation tensor per BN layer.
* ``MfmaResNet``: float32 ResNet trunk on hand-written kernels only -- the stem (``csrc/stem_mfma.hip``: uint8 or float32
  patches -> conv7x7 + bias + ReLU + max-pool, one kernel) and every block convolution (``csrc/conv_mfma.hip``: implicit GEMM
  on the matrix cores with bias / residual / ReLU in the epilogue).
  In fp16 / bf16 the same module runs on ``tia_stem_conv7x7_pool_nhwc_h`` / ``tia_conv2d_nhwc_h`` (no library convolution).

State-dict compatibility is untouched: these are derived copies built from a loaded ``CNNModel`` (reference parameter
names), never the object that loads weights.  Every wrapper raises on tensors it cannot take (host tensors, wrong layout):
there is no silent torch fallback.
"""

from __future__ import annotations

import copy

import torch
import torch.nn.functional as F  # noqa: N812
from torch import nn
from torch.nn.utils.fusion import fuse_conv_bn_eval

from tiatoolbox_amd.models.architecture.resnet import BasicBlock, Bottleneck


def fold_conv_bn(model: nn.Module) -> nn.Module:
    """Deep-copied eval model with every (Conv2d, BatchNorm2d) pair fused."""
    model = copy.deepcopy(model).eval()

    def visit(mod: nn.Module) -> None:
        prev_name, prev = None, None
        for name, child in list(mod.named_children()):
            if isinstance(child, nn.BatchNorm2d) and isinstance(prev, nn.Conv2d):
                setattr(mod, prev_name, fuse_conv_bn_eval(prev, child))
                setattr(mod, name, nn.Identity())
                prev_name, prev = None, None
                continue
            visit(child)
            prev_name, prev = name, child

    visit(model)
    return model


# ------------------------------------------------------------------------------------------------
# HIP epilogues: conv (MIOpen, no bias) -> one fused bias (+ residual) + ReLU pass (cnn_epilogue.hip)
# ------------------------------------------------------------------------------------------------
_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def _nhwc_ptr_ok(t: torch.Tensor) -> bool:
    return t.is_cuda and t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last) and t.dtype in _DT


def hip_bias_act_(x: torch.Tensor, bias: torch.Tensor, residual: torch.Tensor | None = None, *, relu: bool = True) -> torch.Tensor:
    """In place ``x = relu(x + bias[c] (+ residual))`` on an NCHW tensor stored channels-last."""
    from tiatoolbox_amd import _lib

    if not _nhwc_ptr_ok(x) or (residual is not None and not _nhwc_ptr_ok(residual)):
        msg = "hip_bias_act_ expects channels-last CUDA tensors (fp32/fp16/bf16)."
        raise ValueError(msg)
    n, c, h, w = x.shape
    with torch.cuda.device(x.device):
        rc = _lib.load().tia_bias_act_nhwc(x.data_ptr(), bias.data_ptr(), residual.data_ptr() if residual is not None else 0,
                                           n * h * w, c, _DT[x.dtype], int(relu), _lib.current_stream())
    _lib.check(rc, "tia_bias_act_nhwc")
    return x


def hip_bias_relu_maxpool(x: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """``maxpool3x3/s2/p1(relu(x + bias))`` of a channels-last tensor, one pass."""
    from tiatoolbox_amd import _lib

    if not _nhwc_ptr_ok(x):
        msg = "hip_bias_relu_maxpool expects a channels-last CUDA tensor (fp32/fp16/bf16)."
        raise ValueError(msg)
    n, c, h, w = x.shape
    out = torch.empty((n, c, (h + 1) // 2, (w + 1) // 2), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = _lib.load().tia_bias_relu_maxpool_nhwc(x.data_ptr(), bias.data_ptr(), n, h, w, c, _DT[x.dtype], out.data_ptr(),
                                                    _lib.current_stream())
    _lib.check(rc, "tia_bias_relu_maxpool_nhwc")
    return out


# ------------------------------------------------------------------------------------------------
# Hand-written MFMA convolutions (conv_mfma.hip): float32 implicit GEMM with the epilogue fused in
# ------------------------------------------------------------------------------------------------
def hip_conv2d(x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor | None, residual: torch.Tensor | None, *,
               kernel: int, stride: int, padding: int, relu: bool) -> torch.Tensor:
    """``relu(conv2d(x, w) + bias + residual)`` on a float32 channels-last CUDA tensor (``tia_conv2d_nhwc_f32``)."""
    from tiatoolbox_amd import _lib

    if not (_nhwc_ptr_ok(x) and x.dtype == torch.float32):
        msg = "hip_conv2d expects a float32 channels-last CUDA tensor."
        raise ValueError(msg)
    n, cin, h, w = x.shape
    cout = w_packed.shape[-1]
    ho = (h + 2 * padding - kernel) // stride + 1
    wo = (w + 2 * padding - kernel) // stride + 1
    y = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = _lib.load().tia_conv2d_nhwc_f32(x.data_ptr(), w_packed.data_ptr(), bias.data_ptr() if bias is not None else 0,
                                             residual.data_ptr() if residual is not None else 0, y.data_ptr(), n, h, w, cin,
                                             cout, kernel, kernel, stride, padding, int(relu), _lib.current_stream())
    _lib.check(rc, "tia_conv2d_nhwc_f32")
    return y


def hip_conv2d_ex(x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor | None, residual: torch.Tensor | None, *,
                  kernel: int, stride: int, pad_lo: int, pad_hi: int, relu: bool) -> torch.Tensor:
    """Like :func:`hip_conv2d` with ``pad_lo`` zero rows / columns in front and ``pad_hi`` behind (``tia_conv2d_nhwc_f32_ex``):
    TensorFlow-style "same" padding of strided convolutions, and valid convolutions (0 / 0)."""
    from tiatoolbox_amd import _lib

    if not (_nhwc_ptr_ok(x) and x.dtype == torch.float32):
        msg = "hip_conv2d_ex expects a float32 channels-last CUDA tensor."
        raise ValueError(msg)
    if residual is not None and not (_nhwc_ptr_ok(residual) and residual.dtype == torch.float32):
        msg = "hip_conv2d_ex expects a float32 channels-last CUDA residual."
        raise ValueError(msg)
    n, cin, h, w = x.shape
    cout = w_packed.shape[-1]
    ho = (h + pad_lo + pad_hi - kernel) // stride + 1
    wo = (w + pad_lo + pad_hi - kernel) // stride + 1
    y = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    if residual is not None and residual.shape != y.shape:
        msg = f"residual shape {tuple(residual.shape)} != output shape {tuple(y.shape)}"
        raise ValueError(msg)
    with torch.cuda.device(x.device):
        rc = _lib.load().tia_conv2d_nhwc_f32_ex(x.data_ptr(), w_packed.data_ptr(), bias.data_ptr() if bias is not None else 0,
                                                residual.data_ptr() if residual is not None else 0, y.data_ptr(), n, h, w, cin,
                                                cout, kernel, kernel, stride, pad_lo, pad_lo, ho, wo, int(relu),
                                                _lib.current_stream())
    _lib.check(rc, "tia_conv2d_nhwc_f32_ex")
    return y


def pack_thin_conv_weights(weight: torch.Tensor) -> torch.Tensor:
    """OIHW ``[cout, c, kh, kw]`` with ``c * kw <= 32`` -> ``[kh][32][cout]`` (row ``kx * c + ch``; zero rows behind): the
    row-packed form :func:`hip_conv2d_thin` multiplies with."""
    cout, c, kh, kw = weight.shape
    if c * kw > 32 or cout % 64 != 0:  # noqa: PLR2004
        msg = f"thin-input convolution needs c * kw <= 32 and cout % 64 == 0; got weight {tuple(weight.shape)}."
        raise ValueError(msg)
    rows = weight.detach().to(torch.float32).permute(2, 3, 1, 0).reshape(kh, kw * c, cout)
    packed = torch.zeros((kh, 32, cout), dtype=torch.float32, device=weight.device)
    packed[:, :kw * c] = rows
    return packed.contiguous()


def hip_conv2d_thin(x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor | None, *, kernel: int, stride: int, pad_lo: int,
                    pad_hi: int, relu: bool) -> torch.Tensor:
    """Convolution of a few-channel image (``c * kernel <= 32``: HoVer-Net's RGB 7x7 stem) on the MFMA kernel
    (``tia_conv2d_thin_nhwc_f32``): the ``kernel * c`` values under a row of taps are contiguous in NHWC, so they are read
    as one 32-wide slice.  ``x``: float32 channels-last ``[n, c, h, w]``; the horizontal padding (``pad_lo`` / ``pad_hi`` zero
    columns, plus the few that keep the last slice inside its row) is materialised here, the vertical one is the kernel's."""
    from tiatoolbox_amd import _lib

    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4):  # noqa: PLR2004
        msg = "hip_conv2d_thin expects a float32 CUDA tensor [n, c, h, w]."
        raise ValueError(msg)
    n, c, h, w = x.shape
    cout = w_packed.shape[-1]
    ho = (h + pad_lo + pad_hi - kernel) // stride + 1
    wo = (w + pad_lo + pad_hi - kernel) // stride + 1
    need = (wo - 1) * stride + -(-32 // c)  # columns the last output's 32-float read covers
    extra = max(need - (w + pad_lo + pad_hi), 0)
    rows = x.permute(0, 2, 3, 1)  # NHWC
    wp = w + pad_lo + pad_hi + extra
    xp = torch.zeros((n, h, wp, c), dtype=torch.float32, device=x.device)
    xp[:, :, pad_lo:pad_lo + w] = rows
    y = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = _lib.load().tia_conv2d_thin_nhwc_f32(xp.data_ptr(), w_packed.data_ptr(), bias.data_ptr() if bias is not None else 0,
                                                  y.data_ptr(), n, h, wp, c, cout, kernel, kernel, stride, pad_lo, ho, wo, int(relu),
                                                  _lib.current_stream())
    _lib.check(rc, "tia_conv2d_thin_nhwc_f32")
    return y


def hip_conv1x1_head(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None, *, pre_scale: torch.Tensor | None = None,
                     pre_shift: torch.Tensor | None = None) -> torch.Tensor:
    """A class head: 1x1 convolution ``64 -> cout <= 8`` (``tia_conv1x1_head_nhwc_f32``), optionally of
    ``relu(x * pre_scale[c] + pre_shift[c])`` (the BatchNorm + ReLU in front of HoVer-Net's ``u0/conv``) without writing
    that intermediate.  ``x``: float32 channels-last ``[n, 64, h, w]``; ``weight``: ``[cout, 64]`` (or OIHW 1x1)."""
    from tiatoolbox_amd import _lib

    if not (_nhwc_ptr_ok(x) and x.dtype == torch.float32 and x.shape[1] == 64):  # noqa: PLR2004
        msg = "hip_conv1x1_head expects a float32 channels-last CUDA tensor with 64 channels."
        raise ValueError(msg)
    n, _, h, w = x.shape
    cout = weight.shape[0]
    wmat = weight.detach().reshape(cout, 64).to(torch.float32).contiguous()
    y = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = _lib.load().tia_conv1x1_head_nhwc_f32(x.data_ptr(), n * h * w, wmat.data_ptr(), bias.data_ptr() if bias is not None else 0,
                                                   pre_scale.data_ptr() if pre_scale is not None else 0,
                                                   pre_shift.data_ptr() if pre_shift is not None else 0, cout, y.data_ptr(),
                                                   _lib.current_stream())
    _lib.check(rc, "tia_conv1x1_head_nhwc_f32")
    return y


def hip_conv2d_post(x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor | None, residual: torch.Tensor | None, *,
                    kernel: int, stride: int, pad_lo: int, pad_hi: int, relu: bool, post_scale: torch.Tensor,
                    post_shift: torch.Tensor, want_raw: bool = True) -> tuple[torch.Tensor | None, torch.Tensor]:
    """:func:`hip_conv2d_ex` plus ``relu(v * post_scale[c] + post_shift[c])`` of its result ``v`` from the same epilogue
    (``tia_conv2d_post_nhwc_f32``); returns ``(v or None, activated)``."""
    from tiatoolbox_amd import _lib

    if not (_nhwc_ptr_ok(x) and x.dtype == torch.float32):
        msg = "hip_conv2d_post expects a float32 channels-last CUDA tensor."
        raise ValueError(msg)
    n, cin, h, w = x.shape
    cout = w_packed.shape[-1]
    ho = (h + pad_lo + pad_hi - kernel) // stride + 1
    wo = (w + pad_lo + pad_hi - kernel) // stride + 1
    shape = (n, cout, ho, wo)
    if residual is not None and not (_nhwc_ptr_ok(residual) and residual.dtype == torch.float32 and residual.shape == shape):
        msg = "hip_conv2d_post: residual must be a float32 channels-last CUDA tensor of the output shape."
        raise ValueError(msg)
    y = torch.empty(shape, dtype=torch.float32, device=x.device, memory_format=torch.channels_last) if want_raw else None
    y2 = torch.empty(shape, dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = _lib.load().tia_conv2d_post_nhwc_f32(x.data_ptr(), w_packed.data_ptr(), bias.data_ptr() if bias is not None else 0,
                                                  residual.data_ptr() if residual is not None else 0,
                                                  y.data_ptr() if y is not None else 0, n, h, w, cin, cout, kernel, kernel, stride,
                                                  pad_lo, pad_lo, ho, wo, int(relu), post_scale.data_ptr(), post_shift.data_ptr(),
                                                  y2.data_ptr(), _lib.current_stream())
    _lib.check(rc, "tia_conv2d_post_nhwc_f32")
    return y, y2


def hip_conv1x1_pre(x: torch.Tensor, pre_scale: torch.Tensor, pre_shift: torch.Tensor, w_packed: torch.Tensor,
                    bias: torch.Tensor | None, residual: torch.Tensor | None = None, *, stride: int = 1,
                    relu: bool = False) -> torch.Tensor:
    """``act(conv1x1(relu(x * pre_scale[c] + pre_shift[c])) + bias [+ residual])`` with the activation applied on load
    (``tia_conv1x1_pre_nhwc_f32``): the pre-activation in front of a residual unit without an activated copy in memory."""
    from tiatoolbox_amd import _lib

    if not (_nhwc_ptr_ok(x) and x.dtype == torch.float32):
        msg = "hip_conv1x1_pre expects a float32 channels-last CUDA tensor."
        raise ValueError(msg)
    n, cin, h, w = x.shape
    cout = w_packed.shape[-1]
    shape = (n, cout, (h - 1) // stride + 1, (w - 1) // stride + 1)
    if residual is not None and not (_nhwc_ptr_ok(residual) and residual.dtype == torch.float32 and residual.shape == shape):
        msg = "hip_conv1x1_pre: residual must be a float32 channels-last CUDA tensor of the output shape."
        raise ValueError(msg)
    y = torch.empty(shape, dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = _lib.load().tia_conv1x1_pre_nhwc_f32(x.data_ptr(), pre_scale.data_ptr(), pre_shift.data_ptr(), w_packed.data_ptr(),
                                                  bias.data_ptr() if bias is not None else 0,
                                                  residual.data_ptr() if residual is not None else 0, y.data_ptr(), n, h, w, cin,
                                                  cout, stride, int(relu), _lib.current_stream())
    _lib.check(rc, "tia_conv1x1_pre_nhwc_f32")
    return y


def hip_scale_shift_act(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, *, relu: bool = True,
                        inplace: bool = False) -> torch.Tensor:
    """``relu(x * scale[c] + shift[c])`` on a float32 channels-last CUDA tensor (``tia_scale_shift_act_nhwc_f32``)."""
    from tiatoolbox_amd import _lib

    if not (_nhwc_ptr_ok(x) and x.dtype == torch.float32):
        msg = "hip_scale_shift_act expects a float32 channels-last CUDA tensor."
        raise ValueError(msg)
    n, c, h, w = x.shape
    y = x if inplace else torch.empty_like(x, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = _lib.load().tia_scale_shift_act_nhwc_f32(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(), n * h * w, c,
                                                      int(relu), _lib.current_stream())
    _lib.check(rc, "tia_scale_shift_act_nhwc_f32")
    return y


def hip_scale_shift_act_view(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, *, relu: bool = True) -> torch.Tensor:
    """``relu(x * scale[c] + shift[c])`` of a channel-prefix / spatial-window VIEW of a channels-last float32 CUDA tensor,
    written to a dense channels-last tensor (``tia_scale_shift_act_view_nhwc_f32``)."""
    from tiatoolbox_amd import _lib

    if not x.is_cuda:
        msg = "hip_scale_shift_act_view expects a CUDA tensor."
        raise ValueError(msg)
    n, c, h, w = x.shape
    ok = (x.is_cuda and x.dtype == torch.float32 and x.stride(1) == 1 and c % 4 == 0 and x.data_ptr() % 16 == 0
          and all(x.stride(d) % 4 == 0 for d in (0, 2, 3)) and x.stride(3) >= c)
    if not ok:
        msg = ("hip_scale_shift_act_view expects a float32 view with contiguous channels, c % 4 == 0, a 16-byte aligned base and "
               f"strides that are multiples of 4 elements; got shape {tuple(x.shape)} strides {tuple(x.stride())} {x.dtype}.")
        raise ValueError(msg)
    y = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = _lib.load().tia_scale_shift_act_view_nhwc_f32(x.data_ptr(), x.stride(0), x.stride(2), x.stride(3), scale.data_ptr(),
                                                           shift.data_ptr(), y.data_ptr(), n, h, w, c, int(relu),
                                                           _lib.current_stream())
    _lib.check(rc, "tia_scale_shift_act_view_nhwc_f32")
    return y


def hip_grouped_conv_valid(x: torch.Tensor, w_packed: torch.Tensor, *, groups: int, kernel: int,
                           out: torch.Tensor | None = None) -> torch.Tensor:
    """Grouped valid ``kernel x kernel`` convolution, 32 -> 8 channels per group (``tia_grouped_conv_valid_nhwc_f32``); ``out``
    may be a channel slice / window view of a wider channels-last buffer.  ``w_packed``: ``[groups, k, k, 32, 8]``."""
    from tiatoolbox_amd import _lib

    if not (_nhwc_ptr_ok(x) and x.dtype == torch.float32):
        msg = "hip_grouped_conv_valid expects a float32 channels-last CUDA tensor."
        raise ValueError(msg)
    n, cin, h, w = x.shape
    ho, wo = h - kernel + 1, w - kernel + 1
    if out is None:
        out = torch.empty((n, groups * 8, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    if (out.shape != (n, groups * 8, ho, wo) or out.stride(1) != 1 or out.dtype != torch.float32 or out.data_ptr() % 16
            or any(out.stride(d) % 4 for d in (0, 2, 3))):
        msg = "hip_grouped_conv_valid: `out` must be a float32 channels-last (view of a) tensor of the output shape."
        raise ValueError(msg)
    with torch.cuda.device(x.device):
        rc = _lib.load().tia_grouped_conv_valid_nhwc_f32(x.data_ptr(), w_packed.data_ptr(), out.data_ptr(), out.stride(0),
                                                         out.stride(2), out.stride(3), n, h, w, groups, cin // groups, 8, kernel,
                                                         _lib.current_stream())
    _lib.check(rc, "tia_grouped_conv_valid_nhwc_f32")
    return out


def hip_upsample2x_add(x: torch.Tensor, y: torch.Tensor, scale: torch.Tensor | None = None,
                       shift: torch.Tensor | None = None) -> torch.Tensor:
    """``x.repeat_interleave(2, 2).repeat_interleave(2, 3) + y`` in one pass (``tia_upsample2x_add_act_nhwc_f32``); ``y`` may be
    a centre-cropped view of a channels-last tensor.  With ``scale`` / ``shift``: followed by ``relu(. * scale + shift)``."""
    from tiatoolbox_amd import _lib

    if not (x.is_cuda and y.is_cuda):
        msg = "hip_upsample2x_add expects CUDA tensors."
        raise ValueError(msg)
    n, c, h, w = x.shape
    ok_y = (y.is_cuda and y.dtype == torch.float32 and y.shape == (n, c, 2 * h, 2 * w) and y.stride(1) == 1 and y.stride(3) == c
            and y.stride(2) % 4 == 0 and y.stride(0) % 4 == 0 and y.data_ptr() % 16 == 0)
    if not (_nhwc_ptr_ok(x) and x.dtype == torch.float32 and ok_y and c % 4 == 0):
        msg = ("hip_upsample2x_add expects float32 channels-last tensors, `y` a (cropped) view with contiguous channels of shape "
               f"[n, c, 2h, 2w], c % 4 == 0; got x {tuple(x.shape)} {x.dtype}, y {tuple(y.shape)} strides {tuple(y.stride())}.")
        raise ValueError(msg)
    out = torch.empty((n, c, 2 * h, 2 * w), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = _lib.load().tia_upsample2x_add_act_nhwc_f32(x.data_ptr(), y.data_ptr(), y.stride(0), y.stride(2),
                                                         scale.data_ptr() if scale is not None else 0,
                                                         shift.data_ptr() if shift is not None else 0, out.data_ptr(), n, h, w, c,
                                                         _lib.current_stream())
    _lib.check(rc, "tia_upsample2x_add_act_nhwc_f32")
    return out


def hip_conv2d_h(x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor | None, residual: torch.Tensor | None, *,
                 cout: int, kernel: int, stride: int, padding: int, relu: bool) -> torch.Tensor:
    """``relu(conv2d(x, w) + bias + residual)`` on an fp16 / bf16 channels-last CUDA tensor (``tia_conv2d_nhwc_h``: MFMA with
    float32 accumulation; ``bias`` float32; one rounding to half at the end)."""
    from tiatoolbox_amd import _lib

    if not (_nhwc_ptr_ok(x) and x.dtype in (torch.float16, torch.bfloat16)):
        msg = "hip_conv2d_h expects an fp16 / bf16 channels-last CUDA tensor."
        raise ValueError(msg)
    n, cin, h, w = x.shape
    ho = (h + 2 * padding - kernel) // stride + 1
    wo = (w + 2 * padding - kernel) // stride + 1
    y = torch.empty((n, cout, ho, wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    if residual is not None and not (_nhwc_ptr_ok(residual) and residual.dtype == x.dtype and residual.shape == y.shape):
        msg = "hip_conv2d_h: residual must be a channels-last CUDA tensor of the output's dtype and shape."
        raise ValueError(msg)
    if bias is not None and bias.dtype != torch.float32:
        msg = "hip_conv2d_h takes the bias in float32."
        raise ValueError(msg)
    with torch.cuda.device(x.device):
        rc = _lib.load().tia_conv2d_nhwc_h(x.data_ptr(), w_packed.data_ptr(), bias.data_ptr() if bias is not None else 0,
                                           residual.data_ptr() if residual is not None else 0, y.data_ptr(), n, h, w, cin, cout,
                                           kernel, kernel, stride, padding, _DT[x.dtype], int(relu), _lib.current_stream())
    _lib.check(rc, "tia_conv2d_nhwc_h")
    return y


def pack_conv_weights_h(conv: nn.Conv2d, dtype: torch.dtype) -> torch.Tensor:
    """OIHW -> ``[kh, kw, cin/8, cout, 8]`` halves of ``dtype`` on the convolution's device (``tia_conv_pack_weights_h``)."""
    from tiatoolbox_amd import _lib

    w = conv.weight.detach().to(torch.float32).contiguous()
    cout, cin, kh, kw = w.shape
    out = torch.empty((kh, kw, cin // 8, cout, 8), dtype=dtype, device=w.device)
    with torch.cuda.device(w.device):
        rc = _lib.load().tia_conv_pack_weights_h(w.data_ptr(), cout, cin, kh, kw, _DT[dtype], out.data_ptr(), _lib.current_stream())
    _lib.check(rc, "tia_conv_pack_weights_h")
    return out


def pack_conv_weights(conv: nn.Conv2d) -> torch.Tensor:
    """OIHW -> ``[kh, kw, cin, cout]`` float32 on the convolution's device (``tia_conv_pack_weights_f32``)."""
    from tiatoolbox_amd import _lib

    w = conv.weight.detach().to(torch.float32).contiguous()
    cout, cin, kh, kw = w.shape
    out = torch.empty((kh, kw, cin, cout), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        rc = _lib.load().tia_conv_pack_weights_f32(w.data_ptr(), cout, cin, kh, kw, out.data_ptr(), _lib.current_stream())
    _lib.check(rc, "tia_conv_pack_weights_f32")
    return out


def pack_conv_weights_wino(conv: nn.Conv2d) -> torch.Tensor:
    """OIHW 3x3 float32 -> the Winograd-domain weights ``U = G g G^T`` in the stage layout of ``tia_conv3x3_wino_nhwc_f32``
    (``tia_conv_pack_weights_wino_f32``: float64 transform, one rounding), ``[16, cin/16, 2, cout/64, 2, 64, 4]``
    (position, 16-channel slice, 8-channel half, 64-column block, 4-channel group, column, channel)."""
    from tiatoolbox_amd import _lib

    w = conv.weight.detach().to(torch.float32).contiguous()
    cout, cin, kh, kw = w.shape
    if (kh, kw) != (3, 3) or cin % 16 or cout % 64:
        msg = f"Winograd F(2x2, 3x3) needs a 3x3 kernel, cin % 16 == 0 and cout % 64 == 0; got weight {tuple(w.shape)}."
        raise ValueError(msg)
    out = torch.empty((16, cin // 16, 2, cout // 64, 2, 64, 4), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        rc = _lib.load().tia_conv_pack_weights_wino_f32(w.data_ptr(), cout, cin, out.data_ptr(), _lib.current_stream())
    _lib.check(rc, "tia_conv_pack_weights_wino_f32")
    return out


def hip_conv3x3_wino(x: torch.Tensor, u_packed: torch.Tensor, bias: torch.Tensor | None, residual: torch.Tensor | None, *,
                     padding: int, relu: bool, pad_hi: int | None = None) -> torch.Tensor:
    """``relu(conv3x3(x, w) + bias + residual)``, stride 1, through the Winograd F(2x2, 3x3) kernel (``tia_conv3x3_wino_nhwc_f32``):
    float32 in / float32 accumulate like :func:`hip_conv2d`, 2.25 x fewer multiplies, results within ~1e-5 (relative) of it."""
    from tiatoolbox_amd import _lib

    if not (_nhwc_ptr_ok(x) and x.dtype == torch.float32):
        msg = "hip_conv3x3_wino expects a float32 channels-last CUDA tensor."
        raise ValueError(msg)
    if residual is not None and not (_nhwc_ptr_ok(residual) and residual.dtype == torch.float32):
        msg = "hip_conv3x3_wino expects a float32 channels-last CUDA residual."
        raise ValueError(msg)
    n, cin, h, w = x.shape
    if (u_packed.dim() != 7 or cin % 16 or tuple(u_packed.shape) != (16, cin // 16, 2, u_packed.shape[3], 2, 64, 4)  # noqa: PLR2004
            or u_packed.dtype != torch.float32 or not u_packed.is_contiguous() or u_packed.device != x.device):
        msg = (f"hip_conv3x3_wino: packed weights {tuple(u_packed.shape)} {u_packed.dtype} on {u_packed.device} do not match an input with "
               f"{cin} channels on {x.device} (expected pack_conv_weights_wino's [16, cin/16, 2, cout/64, 2, 64, 4] float32, contiguous).")
        raise ValueError(msg)
    cout = u_packed.shape[3] * 64
    behind = padding if pad_hi is None else pad_hi  # zero rows / columns behind the image (`padding` in front): "same", valid, TF-same
    ho, wo = h + padding + behind - 2, w + padding + behind - 2
    y = torch.empty((n, cout, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    if residual is not None and residual.shape != y.shape:
        msg = f"residual shape {tuple(residual.shape)} != output shape {tuple(y.shape)}"
        raise ValueError(msg)
    with torch.cuda.device(x.device):
        rc = _lib.load().tia_conv3x3_wino_nhwc_f32(x.data_ptr(), u_packed.data_ptr(), bias.data_ptr() if bias is not None else 0,
                                                   residual.data_ptr() if residual is not None else 0, y.data_ptr(), n, h, w, cin, cout,
                                                   padding, padding, ho, wo, int(relu), _lib.current_stream())
    _lib.check(rc, "tia_conv3x3_wino_nhwc_f32")
    return y


class _MfmaBlock(nn.Module):
    """Shared machinery of the hand-written blocks: per-convolution packed weights (float32: ``[kh, kw, cin, cout]``; fp16 /
    bf16: ``[kh, kw, cin/8, cout, 8]``), float32 biases, and one launch per convolution with its epilogue fused."""

    conv_algo = "direct"  # "winograd": float32 3x3 / stride-1 layers through tia_conv3x3_wino_nhwc_f32 (MfmaResNet.set_conv_algo)

    def __init__(self) -> None:
        super().__init__()
        self._packed: dict[tuple[str, torch.dtype], torch.Tensor] = {}
        self._bias32: dict[str, torch.Tensor] = {}

    def _wino(self, name: str) -> torch.Tensor | None:
        """The layer's Winograd-domain weights if it is to run on that kernel (float32 3x3 / stride 1, cin % 16, cout % 64)."""
        conv = getattr(self, name)
        if (self.conv_algo != "winograd" or conv.kernel_size != (3, 3) or conv.stride != (1, 1) or conv.dilation != (1, 1)
                or conv.groups != 1 or conv.in_channels % 16 or conv.out_channels % 64 or conv.padding[0] != conv.padding[1]
                or conv.padding[0] > 2):  # noqa: PLR2004
            return None
        cached = self._packed.get((name, "wino"))
        if cached is None or cached.device != conv.weight.device:
            cached = self._packed[(name, "wino")] = pack_conv_weights_wino(conv)
        return cached

    def _w(self, name: str, dtype: torch.dtype) -> torch.Tensor:
        conv = getattr(self, name)
        cached = self._packed.get((name, dtype))
        if cached is None or cached.device != conv.weight.device:
            cached = pack_conv_weights(conv) if dtype == torch.float32 else pack_conv_weights_h(conv, dtype)
            self._packed[(name, dtype)] = cached
        return cached

    def _b(self, name: str) -> torch.Tensor | None:
        conv = getattr(self, name)
        if conv.bias is None:
            return None
        cached = self._bias32.get(name)  # kept from before a cast to half (`prepare`): the kernels add the bias in float32
        if cached is not None and cached.device == conv.bias.device:
            return cached
        if conv.bias.dtype == torch.float32:
            return conv.bias
        cached = self._bias32[name] = conv.bias.detach().float().contiguous()
        return cached

    def prepare(self, dtype: torch.dtype) -> None:
        """Pack the weights for ``dtype`` and keep float32 biases NOW, from the float32 parameters (call before the module is
        cast to half: packing afterwards would start from weights and biases already rounded to half)."""
        for name in ("conv1", "conv2", "conv3", "down"):
            conv = getattr(self, name, None)
            if conv is None:
                continue
            self._w(name, dtype)
            if dtype == torch.float32:
                self._wino(name)
            if conv.bias is not None:
                self._bias32[name] = conv.bias.detach().float().clone().contiguous()

    def _conv(self, name: str, x: torch.Tensor, residual: torch.Tensor | None, *, relu: bool) -> torch.Tensor:
        conv = getattr(self, name)
        k, st, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        if x.dtype == torch.float32:
            u = self._wino(name)
            if u is not None:
                return hip_conv3x3_wino(x, u, self._b(name), residual, padding=pad, relu=relu)
            return hip_conv2d(x, self._w(name, x.dtype), self._b(name), residual, kernel=k, stride=st, padding=pad, relu=relu)
        return hip_conv2d_h(x, self._w(name, x.dtype), self._b(name), residual, cout=conv.out_channels, kernel=k, stride=st,
                            padding=pad, relu=relu)


class _MfmaBasic(_MfmaBlock):
    """BasicBlock as three launches: (downsample) / conv1+bias+ReLU / conv2+bias+residual+ReLU."""

    def __init__(self, blk: BasicBlock) -> None:
        super().__init__()
        self.conv1, self.conv2 = blk.conv1, blk.conv2
        self.down = blk.downsample[0] if blk.downsample is not None else None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        identity = x if self.down is None else self._conv("down", x, None, relu=False)
        out = self._conv("conv1", x, None, relu=True)
        return self._conv("conv2", out, identity, relu=True)


class _MfmaBottleneck(_MfmaBlock):
    """Bottleneck as (downsample) / 1x1+bias+ReLU / 3x3+bias+ReLU / 1x1+bias+residual+ReLU launches."""

    def __init__(self, blk: Bottleneck) -> None:
        super().__init__()
        self.conv1, self.conv2, self.conv3 = blk.conv1, blk.conv2, blk.conv3
        self.down = blk.downsample[0] if blk.downsample is not None else None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        identity = x if self.down is None else self._conv("down", x, None, relu=False)
        out = self._conv("conv1", x, None, relu=True)
        out = self._conv("conv2", out, None, relu=True)
        return self._conv("conv3", out, identity, relu=True)


def pack_stem_weights(conv) -> torch.Tensor:
    """OIHW ``[64, 3, 7, 7]`` -> the stem GEMM's B matrix ``[148, 64]`` (``(ky, kx, c)`` rows + one zero row).  ``conv``: the
    ``nn.Conv2d`` (stride 2, padding 3 are checked) or its (BN-folded) weight tensor."""
    from tiatoolbox_amd import _lib

    weight = conv.weight if isinstance(conv, nn.Module) else conv
    w = weight.detach().to(torch.float32).contiguous()
    geometry_ok = not isinstance(conv, nn.Conv2d) or (conv.stride == (2, 2) and conv.padding == (3, 3))
    if tuple(w.shape) != (64, 3, 7, 7) or not geometry_ok:
        msg = f"the stem kernel is conv7x7 / stride 2 / pad 3, 3 -> 64 channels; got weight {tuple(w.shape)}."
        raise ValueError(msg)
    out = torch.empty((148, 64), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        rc = _lib.load().tia_stem_pack_weights_f32(w.data_ptr(), out.data_ptr(), _lib.current_stream())
    _lib.check(rc, "tia_stem_pack_weights_f32")
    return out


def pack_stem_weights_h(weight: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """OIHW float32 ``[64, 3, 7, 7]`` -> the half stem's B matrix ``[22, 64, 8]`` (k = 24 ky + 3 kx + c, zero padded to 176)."""
    from tiatoolbox_amd import _lib

    w = weight.detach().to(torch.float32).contiguous()
    if tuple(w.shape) != (64, 3, 7, 7) or dtype not in (torch.float16, torch.bfloat16) or not w.is_cuda:
        msg = f"pack_stem_weights_h expects a CUDA float32 weight [64, 3, 7, 7] and fp16 / bf16; got {tuple(w.shape)}, {dtype}."
        raise ValueError(msg)
    out = torch.empty((22, 64, 8), dtype=dtype, device=w.device)
    with torch.cuda.device(w.device):
        rc = _lib.load().tia_stem_pack_weights_h(w.data_ptr(), _DT[dtype], out.data_ptr(), _lib.current_stream())
    _lib.check(rc, "tia_stem_pack_weights_h")
    return out


def hip_stem_conv_pool_h(x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, *, dtype: torch.dtype) -> torch.Tensor:
    """:func:`hip_stem_conv_pool` on the half matrix cores (``tia_stem_conv7x7_pool_nhwc_h``): ``x / 255`` and the weights
    rounded to ``dtype`` (what ``model.half()`` feeds its first convolution), float32 accumulation, bias + ReLU + max-pool in
    float32, ONE rounding of the pooled result to ``dtype``.  ``x``: NHWC uint8 or float32 CUDA batch."""
    from tiatoolbox_amd import _lib

    if not (x.is_cuda and x.dim() == 4 and x.shape[-1] == 3 and x.is_contiguous() and x.dtype in (torch.uint8, torch.float32)):  # noqa: PLR2004
        msg = "hip_stem_conv_pool_h expects a contiguous NHWC uint8 / float32 CUDA batch with 3 channels."
        raise ValueError(msg)
    if w_packed.dtype != dtype or tuple(w_packed.shape) != (22, 64, 8):
        msg = "hip_stem_conv_pool_h expects weights packed by pack_stem_weights_h for the same dtype."
        raise ValueError(msg)
    n, h, w, _ = x.shape
    hp, wp = ((h - 1) // 2) // 2 + 1, ((w - 1) // 2) // 2 + 1
    y = torch.empty((n, 64, hp, wp), dtype=dtype, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = _lib.load().tia_stem_conv7x7_pool_nhwc_h(x.data_ptr(), int(x.dtype == torch.uint8), w_packed.data_ptr(), bias.data_ptr(),
                                                      y.data_ptr(), _DT[dtype], n, h, w, _lib.current_stream())
    _lib.check(rc, "tia_stem_conv7x7_pool_nhwc_h")
    return y


def hip_stem_conv_pool(x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, *,
                       out_dtype: torch.dtype = torch.float32, return_conv: bool = False):
    """``maxpool3x3/2(relu(conv7x7/2(x) + bias))`` in one kernel (``tia_stem_conv7x7_pool_nhwc``).

    ``x``: NHWC ``[n, h, w, 3]`` contiguous CUDA tensor, ``uint8`` (scaled by 1/255 on load: ``ToTensor``) or ``float32`` (as is).
    Returns the pooled activations as an NCHW tensor stored channels-last (``[n, 64, hp, wp]``) of ``out_dtype`` (float32
    arithmetic; fp16 / bf16 = one rounding at the end, for the half-precision trunk).  ``return_conv=True``: ``(pooled, conv)``
    with ``conv = relu(conv7x7(x) + bias)`` before the pooling (``[n, 64, ho, wo]`` float32, the UNet's first skip)."""
    from tiatoolbox_amd import _lib

    if not (x.is_cuda and x.dim() == 4 and x.shape[-1] == 3 and x.is_contiguous() and x.dtype in (torch.uint8, torch.float32)):
        msg = f"hip_stem_conv_pool expects a contiguous NHWC uint8 / float32 CUDA batch with 3 channels, got {tuple(x.shape)} {x.dtype}."
        raise ValueError(msg)
    n, h, w, _ = x.shape
    hp, wp = ((h - 1) // 2) // 2 + 1, ((w - 1) // 2) // 2 + 1
    y = torch.empty((n, 64, hp, wp), dtype=out_dtype, device=x.device, memory_format=torch.channels_last)
    conv = None
    if return_conv:
        conv = torch.empty((n, 64, (h - 1) // 2 + 1, (w - 1) // 2 + 1), dtype=torch.float32, device=x.device,
                           memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = _lib.load().tia_stem_conv7x7_pool_nhwc(x.data_ptr(), int(x.dtype == torch.uint8), w_packed.data_ptr(), bias.data_ptr(),
                                                    y.data_ptr(), _DT[out_dtype], conv.data_ptr() if conv is not None else 0, n, h, w,
                                                    _lib.current_stream())
    _lib.check(rc, "tia_stem_conv7x7_pool_nhwc")
    return (y, conv) if return_conv else y


class MfmaResNet(nn.Module):
    """ResNet trunk on hand-written kernels only: the stem (7x7 / 3 input channels + bias + ReLU + max-pool, one float32 MFMA
    kernel reading uint8 or float32 patches) and every block convolution (BasicBlock: resnet18/34; Bottleneck: resnet50/101) as
    an MFMA implicit GEMM with its epilogue fused (BN folded, channels-last) -- ``tia_conv2d_nhwc_f32`` for float32 (the
    reference's arithmetic), ``tia_conv2d_nhwc_h`` once the module has been cast to fp16 / bf16 (float32 accumulation; the stem
    is then ``tia_stem_conv7x7_pool_nhwc_h``: half inputs and weights on the half matrix cores, one rounding of the result).  A ``uint8`` input means ``ToTensor`` has been deferred into the stem: the
    kernel divides by 255 while it loads."""

    accepts_uint8 = True

    def __init__(self, trunk: nn.Sequential) -> None:
        super().__init__()
        folded = fold_conv_bn(trunk)
        self.stem = folded[0]
        self._stem_packed: torch.Tensor | None = None
        self._stem_packed_dtype: torch.dtype | None = None
        blocks = []
        for layer in list(folded)[4:]:
            for blk in layer:
                if not isinstance(blk, (BasicBlock, Bottleneck)):
                    msg = "MfmaResNet covers BasicBlock / Bottleneck trunks."
                    raise TypeError(msg)
                blocks.append(_MfmaBasic(blk) if isinstance(blk, BasicBlock) else _MfmaBottleneck(blk))
        self.blocks = nn.Sequential(*blocks)

    def stem_forward(self, x: torch.Tensor) -> torch.Tensor:
        """``x``: NCHW view of an NHWC batch (what ``infer_batch`` passes) or the NHWC batch itself."""
        if x.shape[-1] != 3:  # NCHW view -> the NHWC memory underneath (a copy only if it was not channels-last)
            x = x.permute(0, 2, 3, 1)
        x = x.contiguous()
        if x.dtype not in (torch.uint8, torch.float32):
            x = x.to(torch.float32)
        dtype = self.stem.weight.dtype
        w = self._stem_packed
        if w is None or w.device != self.stem.weight.device or self._stem_packed_dtype != dtype:
            self.prepare_stem(dtype)
            w = self._stem_packed
        if dtype == torch.float32:
            return hip_stem_conv_pool(x, w, self._stem_bias)
        return hip_stem_conv_pool_h(x, w, self._stem_bias, dtype=dtype)  # half matrix cores, float32 accumulate

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.blocks(self.stem_forward(x))

    def set_conv_algo(self, algo: str) -> None:
        """``"direct"`` (the module's own default: float32 implicit GEMM, the reference's operation class AND order of accumulation over
        taps) or ``"winograd"``: the float32 3x3 / stride-1 block convolutions through F(2x2, 3x3) -- float32 in, float32 accumulate,
        2.25 x fewer multiplies, log-probabilities within the engine's smoke tolerance of the direct path (the class of arithmetic
        of the vendor libraries' own solvers for these layers).  The ENGINES choose: their default ``conv_algo="auto"`` sets
        ``"winograd"`` here, ``conv_algo="direct"`` is the audit mode (``EngineABC._inference_model``)."""
        if algo not in ("direct", "winograd"):
            msg = f"conv_algo must be 'direct' or 'winograd', got {algo!r}."
            raise ValueError(msg)
        for blk in self.blocks:
            blk.conv_algo = algo

    def prepare(self, dtype: torch.dtype) -> None:
        """Pack every convolution for ``dtype`` from the float32 parameters (the engine calls this on the device, before it
        casts the inference copy to half)."""
        self.prepare_stem(dtype)
        for blk in self.blocks:
            blk.prepare(dtype)

    def prepare_stem(self, dtype: torch.dtype) -> None:
        weight = self.stem.weight.detach().float()
        self._stem_packed = pack_stem_weights(weight) if dtype == torch.float32 else pack_stem_weights_h(weight, dtype)
        self._stem_packed_dtype = dtype
        self._stem_bias = self.stem.bias.detach().float().clone().contiguous()


def fuse_cnn_model(model: nn.Module, *, epilogue_fusion: bool | str = False) -> nn.Module:
    """Derived inference copy of a ``CNNModel``/``CNNBackbone`` with BN folded into the convolutions.

    ``epilogue_fusion="mfma"`` (GPU, any dtype): stem and block convolutions on the hand-written kernels
    (:class:`MfmaResNet`); ``False``: BN folding only (CPU).
    """
    fused = copy.deepcopy(model).eval()
    trunk = fused.feat_extract
    if isinstance(trunk, nn.Sequential) and len(trunk) == 8 and isinstance(trunk[0], nn.Conv2d):
        if epilogue_fusion == "mfma":
            fused.feat_extract = MfmaResNet(trunk)
        elif epilogue_fusion:
            msg = f"unknown epilogue_fusion {epilogue_fusion!r}: 'mfma' or False."
            raise ValueError(msg)
        else:
            fused.feat_extract = fold_conv_bn(trunk)
    return fused
