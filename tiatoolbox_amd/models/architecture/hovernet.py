"""HoVer-Net (API + parameter names of reference ``tiatoolbox/models/architecture/hovernet.py``).

The network is plain torch (convolutions through MIOpen); module / parameter names reproduce the
reference's ``state_dict`` layout (``conv0./.weight``, ``d0.units.0.conv1/bn.weight``,
``decoder.np.u3.dense.units.0.preact_bna/bn.weight`` ...) so its pretrained ``.pth`` files load with
``strict=True``.  The post-processing (``_proc_np_hv``, ``get_instance_info``, ``postproc``) runs as
HIP kernels over whole batches (``tiatoolbox_amd/csrc/hover_post.hip``).
"""

from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F  # noqa: N812
from torch import nn

from tiatoolbox_amd.models.architecture import _hover_device as hd
from tiatoolbox_amd.models.architecture.utils import centre_crop
from tiatoolbox_amd.models.models_abc import ModelABC


def centre_crop_to_shape(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """Centre-crop NCHW ``x`` to the spatial size of ``y`` (ref. ``architecture/utils.py:142-199``)."""
    dh, dw = x.shape[2] - y.shape[2], x.shape[3] - y.shape[3]
    if dh < 0 or dw < 0:
        msg = f"Height or width of `x` is smaller than `y` {list(x.shape[2:])} vs {list(y.shape[2:])}"
        raise ValueError(msg)
    if dh == 0 and dw == 0:
        return x
    return centre_crop(x, (dh, dw))


class UpSample2x(nn.Module):
    """Nearest-neighbour x2 upsampling (ref. ``architecture/utils.py:202-243``); keeps the
    ``unpool_mat`` buffer so reference state dicts load."""

    def __init__(self) -> None:
        super().__init__()
        self.register_buffer("unpool_mat", torch.ones((2, 2), dtype=torch.float32))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)


class TFSamepaddingLayer(nn.Module):
    """TensorFlow 'SAME' padding in front of a valid convolution (ref. :30-69)."""

    def __init__(self, ksize: int, stride: int) -> None:
        super().__init__()
        self.ksize = ksize
        self.stride = stride

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        rem = x.shape[2] % self.stride
        pad = max(self.ksize - (self.stride if rem == 0 else rem), 0)
        lo = pad // 2
        hi = pad - lo
        return F.pad(x, (lo, hi, lo, hi), "constant", 0)


def _bn(ch: int) -> nn.BatchNorm2d:
    return nn.BatchNorm2d(ch, eps=1e-5)


class DenseBlock(nn.Module):
    """Dense units with valid convolutions; features are centre-cropped before concatenation (ref. :72-156)."""

    def __init__(self, in_ch: int, unit_ksizes: list[int], unit_chs: list[int], unit_count: int, split: int = 1) -> None:
        super().__init__()
        if len(unit_ksizes) != len(unit_chs):
            msg = "Unbalance Unit Info."
            raise ValueError(msg)
        self.nr_unit = unit_count
        self.in_ch = in_ch
        self.units = nn.ModuleList()
        ch = in_ch
        for _ in range(unit_count):
            self.units.append(nn.Sequential(OrderedDict([
                ("preact_bna/bn", _bn(ch)),
                ("preact_bna/relu", nn.ReLU(inplace=True)),
                ("conv1", nn.Conv2d(ch, unit_chs[0], unit_ksizes[0], stride=1, padding=0, bias=False)),
                ("conv1/bn", _bn(unit_chs[0])),
                ("conv1/relu", nn.ReLU(inplace=True)),
                ("conv2", nn.Conv2d(unit_chs[0], unit_chs[1], unit_ksizes[1], groups=split, stride=1, padding=0,
                                    bias=False)),
            ])))
            ch += unit_chs[1]
        self.blk_bna = nn.Sequential(OrderedDict([("bn", _bn(ch)), ("relu", nn.ReLU(inplace=True))]))

    def forward(self, prev_feat: torch.Tensor) -> torch.Tensor:
        for unit in self.units:
            new_feat = unit(prev_feat)
            prev_feat = torch.cat([centre_crop_to_shape(prev_feat, new_feat), new_feat], dim=1)
        return self.blk_bna(prev_feat)


class ResidualBlock(nn.Module):
    """Pre-activation bottleneck units (ref. :159-261)."""

    def __init__(self, in_ch: int, unit_ksizes: list[int], unit_chs: list[int], unit_count: int, stride: int = 1) -> None:
        super().__init__()
        if len(unit_ksizes) != len(unit_chs):
            msg = "Unbalance Unit Info."
            raise ValueError(msg)
        self.nr_unit = unit_count
        self.in_ch = in_ch
        self.units = nn.ModuleList()
        ch = in_ch
        for idx in range(unit_count):
            s = stride if idx == 0 else 1
            layers = [
                ("preact/bn", _bn(ch)),
                ("preact/relu", nn.ReLU(inplace=True)),
                ("conv1", nn.Conv2d(ch, unit_chs[0], unit_ksizes[0], stride=1, padding=0, bias=False)),
                ("conv1/bn", _bn(unit_chs[0])),
                ("conv1/relu", nn.ReLU(inplace=True)),
                ("conv2/pad", TFSamepaddingLayer(ksize=unit_ksizes[1], stride=s)),
                ("conv2", nn.Conv2d(unit_chs[0], unit_chs[1], unit_ksizes[1], stride=s, padding=0, bias=False)),
                ("conv2/bn", _bn(unit_chs[1])),
                ("conv2/relu", nn.ReLU(inplace=True)),
                ("conv3", nn.Conv2d(unit_chs[1], unit_chs[2], unit_ksizes[2], stride=1, padding=0, bias=False)),
            ]
            if idx == 0:  # the first unit has no pre-activation
                layers = layers[2:]
            self.units.append(nn.Sequential(OrderedDict(layers)))
            ch = unit_chs[-1]
        self.shortcut = (nn.Conv2d(in_ch, unit_chs[-1], 1, stride=stride, bias=False)
                         if in_ch != unit_chs[-1] or stride != 1 else None)
        self.blk_bna = nn.Sequential(OrderedDict([("bn", _bn(ch)), ("relu", nn.ReLU(inplace=True))]))

    def forward(self, prev_feat: torch.Tensor) -> torch.Tensor:
        shortcut = prev_feat if self.shortcut is None else self.shortcut(prev_feat)
        for unit in self.units:
            prev_feat = unit(prev_feat) + shortcut
            shortcut = prev_feat
        return self.blk_bna(prev_feat)


def _contour_column(polys: list) -> np.ndarray:
    """Object array with one ``(k, 2)`` int32 polygon per instance (what ``DataFrame.to_numpy`` yields, ref. :913-931)."""
    col = np.empty(len(polys), dtype=object)
    for i, poly in enumerate(polys):
        col[i] = poly
    return col


class HoVerNet(ModelABC):
    """HoVer-Net: pre-act ResNet-50 encoder, three dense decoders (``tp``/``np``/``hv``) (ref. :264-932)."""

    def __init__(self, num_input_channels: int = 3, num_types: int | None = None, mode: str = "original",
                 nuc_type_dict: dict | None = None) -> None:
        super().__init__()
        self.mode = mode
        self.num_types = num_types
        self.nuc_type_dict = nuc_type_dict
        self.tasks = ["nuclei_segmentation"]
        self.class_dict = {self.tasks[0]: nuc_type_dict}
        if mode not in ["original", "fast"]:
            msg = f"Invalid mode {mode} for HoVerNet. Only support `original` or `fast`."
            raise ValueError(msg)
        stem = [("/", nn.Conv2d(num_input_channels, 64, 7, stride=1, padding=0, bias=False)), ("bn", _bn(64)),
                ("relu", nn.ReLU(inplace=True))]
        if mode == "fast":
            stem = [("pad", TFSamepaddingLayer(ksize=7, stride=1)), *stem]
        self.conv0 = nn.Sequential(OrderedDict(stem))
        self.d0 = ResidualBlock(64, [1, 3, 1], [64, 64, 256], 3, stride=1)
        self.d1 = ResidualBlock(256, [1, 3, 1], [128, 128, 512], 4, stride=2)
        self.d2 = ResidualBlock(512, [1, 3, 1], [256, 256, 1024], 6, stride=2)
        self.d3 = ResidualBlock(1024, [1, 3, 1], [512, 512, 2048], 3, stride=2)
        self.conv_bot = nn.Conv2d(2048, 1024, 1, stride=1, padding=0, bias=False)
        ksize = 5 if mode == "original" else 3
        branches = [("np", self._create_decoder_branch(ksize=ksize, out_ch=2)),
                    ("hv", self._create_decoder_branch(ksize=ksize, out_ch=2))]
        if num_types is not None:
            branches = [("tp", self._create_decoder_branch(ksize=ksize, out_ch=num_types)), *branches]
        self.decoder = nn.ModuleDict(OrderedDict(branches))
        self.upsample2x = UpSample2x()

    @staticmethod
    def _create_decoder_branch(out_ch: int = 2, ksize: int = 5) -> nn.Sequential:
        """Decoder: u3/u2 = conv + dense block + 1x1, u1 = same-padded conv, u0 = BN-ReLU-1x1 (ref. :456-500)."""
        u3 = nn.Sequential(OrderedDict([
            ("conva", nn.Conv2d(1024, 256, ksize, stride=1, padding=0, bias=False)),
            ("dense", DenseBlock(256, [1, ksize], [128, 32], 8, split=4)),
            ("convf", nn.Conv2d(512, 512, 1, stride=1, padding=0, bias=False))]))
        u2 = nn.Sequential(OrderedDict([
            ("conva", nn.Conv2d(512, 128, ksize, stride=1, padding=0, bias=False)),
            ("dense", DenseBlock(128, [1, ksize], [128, 32], 4, split=4)),
            ("convf", nn.Conv2d(256, 256, 1, stride=1, padding=0, bias=False))]))
        u1 = nn.Sequential(OrderedDict([
            ("conva/pad", TFSamepaddingLayer(ksize=ksize, stride=1)),
            ("conva", nn.Conv2d(256, 64, ksize, stride=1, padding=0, bias=False))]))
        u0 = nn.Sequential(OrderedDict([
            ("bn", _bn(64)), ("relu", nn.ReLU(inplace=True)),
            ("conv", nn.Conv2d(64, out_ch, 1, stride=1, padding=0, bias=True))]))
        return nn.Sequential(OrderedDict([("u3", u3), ("u2", u2), ("u1", u1), ("u0", u0)]))

    def forward(self, input_tensor: torch.Tensor) -> dict:
        """NCHW 0..255 input -> ``{branch: logits}`` (ref. :405-454)."""
        x = input_tensor / 255.0
        d0 = self.d0(self.conv0(x))
        d1 = self.d1(d0)
        d2 = self.d2(d1)
        d3 = self.conv_bot(self.d3(d2))
        if self.mode == "original":
            d0, d1 = centre_crop(d0, [184, 184]), centre_crop(d1, [72, 72])
        else:
            d0, d1 = centre_crop(d0, [92, 92]), centre_crop(d1, [36, 36])
        out = OrderedDict()
        for name, branch in self.decoder.items():
            u3 = branch[0](self.upsample2x(d3) + d2)
            u2 = branch[1](self.upsample2x(u3) + d1)
            u1 = branch[2](self.upsample2x(u2) + d0)
            out[name] = branch[3](u1)
        return out

    # ----------------------------------------------------------------------------- inference
    @staticmethod
    def infer_batch(model: nn.Module, batch_data, *, device: str):
        """``np`` = softmax[...,1:], ``hv`` raw, ``tp`` = argmax as float32; NHWC (ref. :861-910).

        NumPy in -> NumPy out (reference behaviour); a CUDA tensor in -> CUDA tensors out.
        """
        on_device = isinstance(batch_data, torch.Tensor) and batch_data.is_cuda
        if not isinstance(batch_data, torch.Tensor):
            batch_data = torch.as_tensor(np.asarray(batch_data))
        param = next(model.parameters())
        x = batch_data.to(device).to(param.dtype).permute(0, 3, 1, 2)
        if torch.device(device).type == "cuda":
            x = x.contiguous(memory_format=torch.channels_last)
        model.eval()
        with torch.inference_mode():
            pred = model(x)
            pred = OrderedDict((k, v.permute(0, 2, 3, 1).float().contiguous()) for k, v in pred.items())
            pred["np"] = F.softmax(pred["np"], dim=-1)[..., 1:]
            if "tp" in pred:
                type_map = torch.argmax(F.softmax(pred["tp"], dim=-1), dim=-1, keepdim=True)
                pred["tp"] = type_map.type(torch.float32)
        outs = (pred["np"], pred["hv"], pred["tp"]) if "tp" in pred else (pred["np"], pred["hv"])
        return outs if on_device else tuple(v.cpu().numpy() for v in outs)

    # ------------------------------------------------------------------------ post-processing
    @staticmethod
    def _proc_np_hv(np_map, hv_map, scale_factor: float = 1):
        """Instance map from the NP / HV heads (ref. :502-616) -- HIP kernels.

        Accepts one patch (``H,W,1`` / ``H,W,2``) or a batch (``N,H,W,1`` / ``N,H,W,2``); NumPy or CUDA.
        """
        ksize = int((20 * scale_factor) + 1)
        obj_size = math.ceil(10 * (scale_factor**2))
        as_numpy = not isinstance(np_map, torch.Tensor)
        npt = torch.as_tensor(np.asarray(np_map)) if as_numpy else np_map
        hvt = torch.as_tensor(np.asarray(hv_map)) if as_numpy else hv_map
        single = npt.dim() == 3
        if single:
            npt, hvt = npt[None], hvt[None]
        from tiatoolbox_amd.utils._tensors import default_device

        dev = npt.device if npt.is_cuda else default_device()
        inst, _ = hd.proc_np_hv(npt.to(dev), hvt.to(dev), ksize=ksize, obj_size=obj_size)
        inst = inst[0] if single else inst
        return inst.cpu().numpy() if as_numpy else inst

    @staticmethod
    def get_instance_info(pred_inst, pred_type=None, offset=(0, 0), *, verbose: bool = True) -> dict:  # noqa: ARG004
        """Per-instance ``box`` / ``centroid`` / ``contours`` / ``type`` / ``prob`` (ref. :618-748): statistics and
        border following run on the device, only the dict is assembled on the host."""
        as_numpy = not isinstance(pred_inst, torch.Tensor)
        from tiatoolbox_amd.utils._tensors import default_device

        inst = torch.as_tensor(np.asarray(pred_inst)).to(default_device()) if as_numpy else pred_inst
        inst = inst.to(torch.int32)[None]
        tmap = None
        num_types = 0
        if pred_type is not None:
            tmap = torch.as_tensor(np.asarray(pred_type)) if not isinstance(pred_type, torch.Tensor) else pred_type
            tmap = tmap.reshape(inst.shape).to(inst.device).to(torch.uint8)
            num_types = int(tmap.max()) + 1
        max_inst = int(inst.max())
        stats, types = hd.instance_stats(inst, tmap, max_inst, num_types)
        meta, points = hd.contours(inst, stats, max_inst)
        return hd.info_from_stats(stats[0].cpu().numpy(), types[0].cpu().numpy() if types is not None else None, offset,
                                  meta=meta[0], points=points)

    def postproc(self, raw_maps: list, offset: tuple[int, int] = (0, 0)) -> tuple[dict, ...]:
        """Post-process one patch/tile (ref. :751-859): instance map + instance table."""
        if len(raw_maps) == 3:  # noqa: PLR2004
            np_map, hv_map, tp_map = raw_maps
            if isinstance(tp_map, torch.Tensor):
                tp_map = torch.round(tp_map).to(torch.uint8)
            else:
                tp_map = np.around(tp_map).astype("uint8")
        else:
            tp_map = None
            np_map, hv_map = raw_maps
        pred_inst = HoVerNet._proc_np_hv(np_map, hv_map)
        info = HoVerNet.get_instance_info(pred_inst, tp_map, offset)
        return (self._pack(pred_inst, info),)

    def _pack(self, pred_inst, info: dict) -> dict:
        if not info:
            empty = np.empty(shape=0)
            table = {"box": empty, "centroid": empty, "contours": empty, "prob": empty, "type": empty}
        else:
            table = {
                "box": np.array([v["box"] for v in info.values()]),
                "centroid": np.array([v["centroid"] for v in info.values()]),
                "contours": _contour_column([v["contours"] for v in info.values()]),
                "prob": np.array([v["prob"] for v in info.values()], dtype=object),
                "type": np.array([v["type"] for v in info.values()], dtype=object),
            }
        return {"task_type": self.tasks[0], "predictions": pred_inst, "info_dict": table, "seg_type": "instance"}

    def postproc_batch(self, np_map: torch.Tensor, hv_map: torch.Tensor, tp_map: torch.Tensor | None) -> list[dict]:
        """Batched device post-processing used by the engines: one launch sequence for all patches."""
        inst, nmark = hd.proc_np_hv(np_map, hv_map)
        tmap, num_types = None, 0
        if tp_map is not None:
            tmap = torch.round(tp_map).to(torch.uint8).reshape(inst.shape)
            num_types = max(int(self.num_types or 0), int(tmap.max()) + 1)
        max_inst = int(nmark.max())
        stats, types = hd.instance_stats(inst, tmap, max_inst, num_types)
        meta, points = hd.contours(inst, stats, max_inst)
        stats_h = stats.cpu().numpy()
        types_h = types.cpu().numpy() if types is not None else None
        inst_h = inst.cpu().numpy()
        # batch-wide column assembly: one NumPy pass per column over all instances of all planes, then per-plane views
        tables = hd.tables_from_stats_batch(stats_h, types_h, meta=meta, points=points)
        return [self._pack_table(inst_h[i], tables[i]) for i in range(inst.shape[0])]

    def _pack_table(self, pred_inst, table: dict | None) -> dict:
        if table is None:
            return self._pack(pred_inst, {})
        cols = {k: table[k] for k in ("box", "centroid", "contours", "prob", "type")}
        return {"task_type": self.tasks[0], "predictions": pred_inst, "info_dict": cols, "seg_type": "instance"}
