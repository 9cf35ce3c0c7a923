"""Inference copy of :class:`HoVerNet` / :class:`HoVerNetPlus` on the hand-written float32 MFMA convolutions.

Same arithmetic graph as ``HoVerNet.forward`` (reference ``models/architecture/hovernet.py:405-454``), re-expressed so
that every convolution with ``cin % 32 == 0`` and ``cout % 64 == 0`` -- all 1x1 / 3x3 convolutions of the
pre-activation ResNet-50 encoder, the decoders' ``conva`` / ``convf`` and the dense units' 1x1 -- runs on
``tia_conv2d_nhwc_f32_ex`` with its epilogue fused:

* ``conv -> BN -> ReLU`` (``conv1`` / ``conv2`` of every unit, the stem): BN folded into the weights, bias + ReLU in the
  convolution's epilogue;
* ``conv3 + shortcut``: the residual add rides in the epilogue of ``conv3``;
* ``BN -> ReLU -> conv1`` (the pre-activation of residual units 2..n): applied to the operand while ``conv1`` loads it
  (``tia_conv1x1_pre_nhwc_f32``) -- the unit reads the previous raw sum, no activated copy exists in memory; ``blk_bna``
  (after the last unit) comes out of ``conv3``'s epilogue as its only output (``tia_conv2d_post_nhwc_f32``); the dense
  units' pre-activations are one ``tia_scale_shift_act_nhwc_f32`` pass;
* TensorFlow "same" padding of the strided 3x3: expressed by the convolution's explicit front padding / output size.

The other convolutions are hand-written too: the 3-channel 7x7 stem runs on the same MFMA kernel in its row-packed
thin-input form (``tia_conv2d_thin_nhwc_f32``), the dense units' grouped convolutions (32 -> 8 channels per group) on
``tia_grouped_conv_valid_nhwc_f32``, and the final ``BN -> ReLU -> 64 -> n_out`` 1x1 of every branch is one launch of
``tia_conv1x1_head_nhwc_f32``.  Built from a loaded model (reference parameter names), never the object that loads
weights; float32, CUDA, channels-last only.
"""

from __future__ import annotations

import os
from collections import OrderedDict

import torch
import torch.nn.functional as F  # noqa: N812
from torch import nn

from tiatoolbox_amd.models.architecture.fused import (hip_bias_act_, hip_conv1x1_head, hip_conv1x1_pre, hip_conv2d_ex, hip_conv2d_post,
                                                      hip_conv3x3_wino, pack_conv_weights_wino,
                                                      hip_conv2d_thin, hip_grouped_conv_valid, hip_scale_shift_act,
                                                      hip_scale_shift_act_view, hip_upsample2x_add, pack_conv_weights,
                                                      pack_thin_conv_weights)
from tiatoolbox_amd.models.architecture.hovernet import centre_crop_to_shape
from tiatoolbox_amd.models.architecture.utils import centre_crop


def _cl(x: torch.Tensor) -> torch.Tensor:
    return x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)


def _bn_affine(bn: nn.BatchNorm2d) -> tuple[torch.Tensor, torch.Tensor]:
    scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach().float().contiguous()
    shift = (bn.bias - bn.running_mean * scale).detach().float().contiguous()
    return scale, shift


def _same_pads(size: int, ksize: int, stride: int) -> tuple[int, int]:
    """``TFSamepaddingLayer`` (ref. :30-69): total pad from the HEIGHT's remainder, smaller half in front."""
    rem = size % stride
    pad = max(ksize - (stride if rem == 0 else rem), 0)
    return pad // 2, pad - pad // 2


class _Conv(nn.Module):
    """One convolution of the graph: weights (optionally with a following BN folded in) packed for the MFMA kernel."""

    def __init__(self, conv: nn.Conv2d, bn: nn.BatchNorm2d | None = None) -> None:
        super().__init__()
        w = conv.weight.detach().float()
        bias = conv.bias.detach().float() if conv.bias is not None else None
        if bn is not None:
            scale, shift = _bn_affine(bn)
            w = w * scale[:, None, None, None]
            bias = shift if bias is None else bias * scale + shift
        self.kernel, self.stride = conv.kernel_size[0], conv.stride[0]
        self.mfma_ok = conv.groups == 1 and conv.in_channels % 32 == 0 and conv.out_channels % 64 == 0
        self.groups = conv.groups
        # few input channels (the RGB stem): row-packed form of the MFMA kernel; few output channels (a class head): the head kernel
        self.thin_ok = (conv.groups == 1 and conv.in_channels < 32 and conv.in_channels * conv.kernel_size[1] <= 32  # noqa: PLR2004
                        and conv.out_channels % 64 == 0 and conv.kernel_size[0] == conv.kernel_size[1])
        self.head_ok = (conv.groups == 1 and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.in_channels == 64  # noqa: PLR2004
                        and conv.out_channels <= 8)  # noqa: PLR2004
        # the dense units' grouped convolution: 32 -> 8 channels per group, stride 1, no bias
        self.grouped_ok = (conv.groups > 1 and conv.in_channels // conv.groups == 32 and conv.out_channels // conv.groups == 8
                           and conv.stride[0] == 1 and bias is None)
        self.weight = nn.Parameter(w.contiguous(), requires_grad=False)
        self.bias = nn.Parameter(bias.contiguous(), requires_grad=False) if bias is not None else None
        self._packed: torch.Tensor | None = None
        # opt-in (`set_conv_algo(model, "winograd")`, engine kwarg `conv_algo`): plain 3x3 / stride-1 layers through the Winograd
        # F(2x2, 3x3) kernel (float32 in / float32 accumulate; csrc/conv3x3_wino.hip)
        self.conv_algo = "direct"
        self.wino_ok = self.mfma_ok and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.dilation == (1, 1)
        self._wino: torch.Tensor | None = None

    def forward(self, x: torch.Tensor, *, pads: tuple[int, int] = (0, 0), relu: bool = False,
                residual: torch.Tensor | None = None, out: torch.Tensor | None = None, pre: "_BnAct | None" = None) -> torch.Tensor:
        if self.head_ok and pads == (0, 0) and not relu and residual is None:
            # `pre`: the BatchNorm + ReLU in front of the head, applied on load
            return hip_conv1x1_head(_cl(x), self.weight, self.bias, pre_scale=pre.scale if pre is not None else None,
                                    pre_shift=pre.shift if pre is not None else None)
        if pre is not None:
            x = pre(x)
        if self.thin_ok and residual is None:
            if self._packed is None or self._packed.device != self.weight.device:
                self._packed = pack_thin_conv_weights(self.weight)
            return hip_conv2d_thin(x, self._packed, self.bias, kernel=self.kernel, stride=self.stride, pad_lo=pads[0], pad_hi=pads[1],
                                   relu=relu)
        if self.grouped_ok and pads == (0, 0) and not relu and residual is None:
            if self._packed is None or self._packed.device != self.weight.device:
                g, k = self.groups, self.kernel
                self._packed = self.weight.view(g, 8, 32, k, k).permute(0, 3, 4, 2, 1).contiguous()  # [g][ky][kx][c][j]
            return hip_grouped_conv_valid(_cl(x), self._packed, groups=self.groups, kernel=self.kernel, out=out)
        if self.wino_ok and self.conv_algo == "winograd" and max(pads) <= 2:  # noqa: PLR2004
            if self._wino is None or self._wino.device != self.weight.device:
                self._wino = pack_conv_weights_wino(self)  # reads `.weight` (OIHW, BN folded)
            return hip_conv3x3_wino(_cl(x), self._wino, self.bias, residual, padding=pads[0], pad_hi=pads[1], relu=relu)
        if self.mfma_ok:
            if self._packed is None or self._packed.device != self.weight.device:
                self._packed = pack_conv_weights(self)  # reads `.weight` (OIHW)
            return hip_conv2d_ex(_cl(x), self._packed, self.bias, residual, kernel=self.kernel, stride=self.stride,
                                 pad_lo=pads[0], pad_hi=pads[1], relu=relu)
        if pads != (0, 0):
            x = F.pad(x, (pads[0], pads[1], pads[0], pads[1]))
        y = _cl(F.conv2d(x, self.weight, None, self.stride, 0, 1, self.groups))
        if self.bias is not None or relu or residual is not None:
            bias = self.bias if self.bias is not None else torch.zeros(y.shape[1], device=y.device)
            if y.shape[1] % 4 == 0:
                return hip_bias_act_(y, bias, residual, relu=relu)
            y = y + bias[None, :, None, None]
            y = y + residual if residual is not None else y
            return F.relu(y) if relu else y
        return y


def set_conv_algo(model: nn.Module, algo: str) -> int:
    """``"direct"`` | ``"winograd"`` for every plain 3x3 / stride-1 MFMA convolution of a fused segmentation network (``FusedHoVerNet``,
    ``FusedUNet``); returns how many layers the switch applies to.  Layers with a second epilogue output or an activation on load
    keep their own kernels."""
    if algo not in ("direct", "winograd"):
        msg = f"conv_algo must be 'direct' or 'winograd', got {algo!r}."
        raise ValueError(msg)
    count = 0
    for mod in model.modules():
        if isinstance(mod, _Conv):
            mod.conv_algo = algo
            count += int(mod.wino_ok)
    return count


def _conv_with_post(conv: "_Conv", x: torch.Tensor, residual: torch.Tensor, bn: "_BnAct", *, want_raw: bool):
    """``v = conv(x) + residual`` and ``relu(bn(v))`` from one launch (1x1 MFMA convolution); ``(v or None, activated)``."""
    if conv._packed is None or conv._packed.device != conv.weight.device:  # noqa: SLF001
        conv._packed = pack_conv_weights(conv)  # noqa: SLF001
    return hip_conv2d_post(_cl(x), conv._packed, conv.bias, residual, kernel=conv.kernel, stride=conv.stride, pad_lo=0,  # noqa: SLF001
                           pad_hi=0, relu=False, post_scale=bn.scale, post_shift=bn.shift, want_raw=want_raw)


def _conv_pre_on_load(conv: "_Conv", x: torch.Tensor, bn: "_BnAct") -> torch.Tensor:
    """``relu(conv(relu(bn(x))) + bias)`` for a 1x1 MFMA convolution, the BN + ReLU applied to the operand on load."""
    if conv._packed is None or conv._packed.device != conv.weight.device:  # noqa: SLF001
        conv._packed = pack_conv_weights(conv)  # noqa: SLF001
    return hip_conv1x1_pre(_cl(x), bn.scale, bn.shift, conv._packed, conv.bias, stride=conv.stride, relu=True)  # noqa: SLF001


# developer switch for A/B measurements: TIA_HOVER_PRE_ON_LOAD=0 keeps the second (activated) epilogue output instead
_PRE_ON_LOAD = os.environ.get("TIA_HOVER_PRE_ON_LOAD", "1") != "0"


class _BnAct(nn.Module):
    def __init__(self, bn: nn.BatchNorm2d) -> None:
        super().__init__()
        scale, shift = _bn_affine(bn)
        self.register_buffer("scale", scale)
        self.register_buffer("shift", shift)

    def forward(self, x: torch.Tensor, *, inplace: bool = False) -> torch.Tensor:
        return hip_scale_shift_act(_cl(x), self.scale, self.shift, relu=True, inplace=inplace)

    def view(self, x: torch.Tensor) -> torch.Tensor:
        """The same for a channel-prefix / window view of a wider channels-last buffer; dense result."""
        return hip_scale_shift_act_view(x, self.scale, self.shift, relu=True)


class _FusedResidualBlock(nn.Module):
    def __init__(self, blk: nn.Module) -> None:
        super().__init__()
        self.pre = nn.ModuleList()
        self.c1, self.c2, self.c3 = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for unit in blk.units:
            mods = dict(unit.named_children())
            self.pre.append(_BnAct(mods["preact/bn"]) if "preact/bn" in mods else nn.Identity())
            self.c1.append(_Conv(mods["conv1"], mods["conv1/bn"]))
            self.c2.append(_Conv(mods["conv2"], mods["conv2/bn"]))
            self.c3.append(_Conv(mods["conv3"]))
        self.shortcut = _Conv(blk.shortcut) if blk.shortcut is not None else None
        self.out = _BnAct(blk.blk_bna.bn)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        shortcut = x if self.shortcut is None else self.shortcut(x)
        units = len(self.c1)
        a = x  # the first unit has no pre-activation
        raw_in = False  # `a` is the raw residual sum: the unit's pre-activation is applied by conv1 on load
        for i, (c1, c2, c3) in enumerate(zip(self.c1, self.c2, self.c3)):
            a = _conv_pre_on_load(c1, a, self.pre[i]) if raw_in else c1(a, relu=True)
            a = c2(a, pads=_same_pads(a.shape[2], c2.kernel, c2.stride), relu=True)
            last = i + 1 == units
            nxt = self.out if last else self.pre[i + 1]
            raw_in = not last and _PRE_ON_LOAD and c3.mfma_ok and self.c1[i + 1].mfma_ok and self.c1[i + 1].kernel == 1
            if raw_in:
                # conv3 + shortcut only: the next unit reads this raw sum twice (conv1 activates it on load, conv3 adds it),
                # so the activated copy is neither written nor read back -- 3 instead of 4 passes over the widest tensor
                a = shortcut = c3(a, residual=_cl(shortcut))
            elif c3.mfma_ok:
                # conv3 + shortcut, and the BN + ReLU that follows the sum (the block's blk_bna, or the next unit's
                # pre-activation when on-load activation is off), from one epilogue
                shortcut, a = _conv_with_post(c3, a, _cl(shortcut), nxt, want_raw=not last)
            else:
                shortcut = c3(a, residual=_cl(shortcut))
                a = nxt(shortcut)
        return a


class _FusedDenseBlock(nn.Module):
    def __init__(self, blk: nn.Module) -> None:
        super().__init__()
        self.pre, self.c1, self.c2 = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for unit in blk.units:
            mods = dict(unit.named_children())
            self.pre.append(_BnAct(mods["preact_bna/bn"]))
            self.c1.append(_Conv(mods["conv1"], mods["conv1/bn"]))
            self.c2.append(_Conv(mods["conv2"]))
        self.out = _BnAct(blk.blk_bna.bn)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # x_{i+1} = cat(centre_crop(x_i), new_i): instead of re-concatenating the growing stack, one buffer holds all
        # channels at the input size; unit i reads the channel prefix / shrinking window it owns and writes its 32 new
        # channels into the next slice (valid k x k convolutions crop (k - 1) / 2 pixels per side and unit)
        n, c0, h0, w0 = x.shape
        units = len(self.c1)
        grow = self.c2[0].weight.shape[0]
        r = (self.c2[0].kernel - 1) // 2
        buf = torch.empty((n, c0 + grow * units, h0, w0), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        buf[:, :c0] = x
        c = c0
        for i, (pre, c1, c2) in enumerate(zip(self.pre, self.c1, self.c2)):
            view = buf[:, :c, i * r:h0 - i * r, i * r:w0 - i * r]
            dst = buf[:, c:c + grow, (i + 1) * r:h0 - (i + 1) * r, (i + 1) * r:w0 - (i + 1) * r]
            if c2.grouped_ok:
                c2(c1(pre.view(view), relu=True), out=dst)  # written straight into its slice
            else:
                dst.copy_(c2(c1(pre.view(view), relu=True)))
            c += grow
        return self.out.view(buf[:, :, units * r:h0 - units * r, units * r:w0 - units * r])


class _FusedBranch(nn.Module):
    def __init__(self, branch: nn.Sequential) -> None:
        super().__init__()
        u3, u2, u1, u0 = branch[0], branch[1], branch[2], branch[3]
        self.u3a, self.u3d, self.u3f = _Conv(u3.conva), _FusedDenseBlock(u3.dense), _Conv(u3.convf)
        self.u2a, self.u2d, self.u2f = _Conv(u2.conva), _FusedDenseBlock(u2.dense), _Conv(u2.convf)
        self.u1a = _Conv(u1.conva)
        self.u1_ksize = u1.conva.kernel_size[0]
        self.u0bn = _BnAct(u0.bn)
        self.u0 = _Conv(u0.conv)


class FusedHoVerNet(nn.Module):
    """``forward(x)`` == ``HoVerNet.forward(x)`` (``{branch: logits}``), float32 on a CUDA device."""

    def __init__(self, model: nn.Module) -> None:
        super().__init__()
        model = model.eval()
        self.mode = model.mode
        stem = dict(model.conv0.named_children())
        self.stem = _Conv(stem["/"], stem["bn"])
        self.stem_pad = "pad" in stem
        self.d0, self.d1 = _FusedResidualBlock(model.d0), _FusedResidualBlock(model.d1)
        self.d2, self.d3 = _FusedResidualBlock(model.d2), _FusedResidualBlock(model.d3)
        self.conv_bot = _Conv(model.conv_bot)
        self.decoder = nn.ModuleDict(OrderedDict((name, _FusedBranch(branch)) for name, branch in model.decoder.items()))

    def forward(self, input_tensor: torch.Tensor) -> dict:
        x = _cl(input_tensor / 255.0)
        pads = _same_pads(x.shape[2], self.stem.kernel, 1) if self.stem_pad else (0, 0)
        d0 = self.d0(self.stem(x, pads=pads, relu=True))
        d1 = self.d1(d0)
        d2 = self.d2(d1)
        d3 = self.conv_bot(self.d3(d2))
        if self.mode == "original":
            d0, d1 = centre_crop(d0, [184, 184]), centre_crop(d1, [72, 72])
        else:
            d0, d1 = centre_crop(d0, [92, 92]), centre_crop(d1, [36, 36])
        up3 = hip_upsample2x_add(_cl(d3), d2)  # shared by the branches
        out = OrderedDict()
        for name, br in self.decoder.items():
            u3 = br.u3f(br.u3d(br.u3a(up3)))
            u2 = br.u2f(br.u2d(br.u2a(hip_upsample2x_add(_cl(u3), d1))))
            u1_in = hip_upsample2x_add(_cl(u2), d0)
            u1 = br.u1a(u1_in, pads=_same_pads(u1_in.shape[2], br.u1_ksize, 1))
            out[name] = br.u0(u1, pre=br.u0bn)  # BN + ReLU + 1x1 head: one launch
        return out
