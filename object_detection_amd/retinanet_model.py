"""Model factory with the signature of the reference's
``object_detection/retinanet/retinanet_model.py:get_model`` (lines 19-61), which
``object_detection/train.py:144`` and ``object_detection/predict.py:44-45`` call.

Scope: this package provides the detection-head box pipeline, not the reference's backbones (timm /
torchvision classifiers with downloaded weights, BiFPN/FPN necks).  ``backbone`` may therefore be
  * an ``nn.Module`` that maps a batch tensor to an ``OrderedDict`` of feature maps and has
    ``.out_channels`` (any of the reference's ``BackboneWithFPN`` / ``BackboneWithBiFPN`` objects fits), or
  * the string ``"pyramid_stub"``: a minimal strided-conv pyramid used for plumbing tests and smoke runs.
A backbone *name* of the reference ('poolformer_s24', 'efficientnet_b4', ...) selects the anchor
geometry the reference would build for it and needs ``backbone_module=`` to supply the network.
Unknown keyword arguments are accepted and ignored, as in the reference call chain (``nms_thres=``).
"""
from __future__ import annotations

from collections import OrderedDict
from functools import partial
from typing import List, Optional

import torch
from torch import nn

from .anchor_utils import AnchorGenerator
from .fcos import FCOS
from .retinanet import RetinaNet, RetinaNetHead


class PyramidStub(nn.Module):
    """Strided 3x3 convs producing ``num_levels`` maps at strides first_stride * 2^i (not a real backbone)."""

    def __init__(self, num_levels: int, out_channels: int = 32, first_stride: int = 4):
        super().__init__()
        self.out_channels = out_channels
        self.stem = nn.Conv2d(3, out_channels, 3, stride=first_stride, padding=1)
        self.down = nn.ModuleList([nn.Conv2d(out_channels, out_channels, 3, stride=2, padding=1)
                                   for _ in range(num_levels - 1)])

    def forward(self, x):
        out = OrderedDict()
        x = torch.relu(self.stem(x))
        out["p0"] = x
        for i, conv in enumerate(self.down):
            x = torch.relu(conv(x))
            out[f"p{i + 1}"] = x
        return out


def anchor_geometry(backbone: str, layers: List[int]):
    """Anchor sizes / ratios exactly as the reference derives them (retinanet_model.py:22-24,32-43)."""
    if backbone == "resnet50":
        base = [16, 32, 64, 128, 256, 512]
    elif "efficientnet" in backbone:
        base = [2 ** (4 + i) for i in range(len(layers))]
    else:   # poolformer / focalnet add P6, P7
        base = [2 ** (4 + i) for i in range(len(layers) + 2)]
    sizes = tuple((x, int(x * 2 ** (1.0 / 3)), int(x * 2 ** (2.0 / 3))) for x in base)
    ratios = ((0.5, 1.0, 2.0),) * len(sizes)
    return sizes, ratios


def get_model(num_classes, pretrained=False, layers=[0, 2, 4, 6], arch="retinanet", backbone="poolformer_s24",
              is_torch=False, out_channels=256, baseline="fpn", num_layers=2,
              backbone_module: Optional[nn.Module] = None, **kwargs):
    if pretrained:
        raise RuntimeError("pretrained weights need a network download, which this build never attempts")
    name = backbone if isinstance(backbone, str) else "custom"
    if isinstance(backbone, nn.Module):
        backbone_module, name = backbone, "custom"
    if arch == "retinanet":
        if name in ("custom", "pyramid_stub"):
            n_levels = len(layers) + 2
            sizes = tuple((x, int(x * 2 ** (1.0 / 3)), int(x * 2 ** (2.0 / 3))) for x in
                          [2 ** (4 + i) for i in range(n_levels)])
            ratios = ((0.5, 1.0, 2.0),) * n_levels
        else:
            sizes, ratios = anchor_geometry(name, layers)
        if backbone_module is None:
            if name != "pyramid_stub":
                raise NotImplementedError(
                    f"backbone '{name}' is outside this package (detection-head box pipeline only); pass the "
                    f"network as backbone_module=<nn.Module with .out_channels> or use backbone='pyramid_stub'")
            backbone_module = PyramidStub(len(sizes), out_channels)
        anchor_generator = AnchorGenerator(sizes, ratios)
        head = RetinaNetHead(backbone_module.out_channels, anchor_generator.num_anchors_per_location()[0],
                             num_classes, norm_layer=partial(nn.GroupNorm, 32))
        head.regression_head._loss_type = "giou"
        return RetinaNet(backbone_module, num_classes=num_classes, anchor_generator=anchor_generator, head=head,
                         **kwargs)
    if arch == "fcos":
        if backbone_module is None:
            if name != "pyramid_stub":
                raise NotImplementedError(f"backbone '{name}' is outside this package; pass backbone_module=")
            backbone_module = PyramidStub(5, out_channels, first_stride=8)
        return FCOS(backbone_module, num_classes, **kwargs)
    raise ValueError(f"unknown arch {arch!r}")
