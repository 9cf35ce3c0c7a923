"""ResNet trunks in plain torch with torchvision-compatible parameter names.

torchvision is not a dependency here; the reference builds its CNN backbones as
``nn.Sequential(*list(torchvision.models.resnetXX().children())[:-2])``
(``tiatoolbox/models/architecture/vanilla.py:157-158``), so pretrained tiatoolbox ``.pth``
files address parameters as ``feat_extract.0.weight`` (conv1), ``feat_extract.1.*`` (bn1),
``feat_extract.4-7.<block>.{conv1,bn1,conv2,bn2,downsample.0,downsample.1}.*``.
The modules below reproduce exactly that layout so reference weights load with
``strict=True``.  Convolutions run through MIOpen / hipBLASLt (MFMA) via PyTorch-ROCm.
"""

from __future__ import annotations

import torch
from torch import nn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: nn.Module | None = None) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        out = out + identity
        return self.relu(out)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: nn.Module | None = None) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        out = out + identity
        return self.relu(out)


_CFG = {
    "resnet18": (BasicBlock, (2, 2, 2, 2)),
    "resnet34": (BasicBlock, (3, 4, 6, 3)),
    "resnet50": (Bottleneck, (3, 4, 6, 3)),
    "resnet101": (Bottleneck, (3, 4, 23, 3)),
}


def _make_layer(block, inplanes: int, planes: int, blocks: int, stride: int) -> tuple[nn.Sequential, int]:
    downsample = None
    if stride != 1 or inplanes != planes * block.expansion:
        downsample = nn.Sequential(
            nn.Conv2d(inplanes, planes * block.expansion, 1, stride, bias=False),
            nn.BatchNorm2d(planes * block.expansion),
        )
    layers = [block(inplanes, planes, stride, downsample)]
    inplanes = planes * block.expansion
    layers += [block(inplanes, planes) for _ in range(1, blocks)]
    return nn.Sequential(*layers), inplanes


def resnet_children(name: str) -> list[nn.Module]:
    """The first eight children of a torchvision ResNet: conv1, bn1, relu, maxpool, layer1..4."""
    if name not in _CFG:
        msg = f"Backbone `{name}` is not supported."
        raise ValueError(msg)
    block, layers = _CFG[name]
    mods: list[nn.Module] = [
        nn.Conv2d(3, 64, 7, 2, 3, bias=False),
        nn.BatchNorm2d(64),
        nn.ReLU(inplace=True),
        nn.MaxPool2d(3, 2, 1),
    ]
    inplanes = 64
    for planes, n, stride in zip((64, 128, 256, 512), layers, (1, 2, 2, 2)):
        layer, inplanes = _make_layer(block, inplanes, planes, n, stride)
        mods.append(layer)
    for m in mods:
        for sub in m.modules():
            if isinstance(sub, nn.Conv2d):
                nn.init.kaiming_normal_(sub.weight, mode="fan_out", nonlinearity="relu")
    return mods


def resnet_trunk(name: str) -> nn.Sequential:
    return nn.Sequential(*resnet_children(name))
