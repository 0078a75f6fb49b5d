"""FCOS with the constructor and method signatures of ``torchvision.models.detection.fcos`` (0.16.2),
which the reference instantiates at ``object_detection/retinanet/retinanet_model.py:15-16,55-60``.
Centre-sampling matching (``compute_loss``) and ``postprocess_detections`` run in HIP kernels.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from functools import partial
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor, nn

from . import _C, det_utils, kernels
from .anchor_utils import AnchorGenerator
from .retinanet import _to_anchor_major, postprocess_batched
from .transform import GeneralizedRCNNTransform


def _gn_tower(in_channels: int, num_convs: int, norm_layer) -> nn.Sequential:
    layers: List[nn.Module] = []
    for _ in range(num_convs):
        layers += [nn.Conv2d(in_channels, in_channels, 3, 1, 1), norm_layer(in_channels), nn.ReLU()]
    tower = nn.Sequential(*layers)
    for m in tower.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.normal_(m.weight, std=0.01)
            nn.init.constant_(m.bias, 0)
    return tower


class FCOSClassificationHead(nn.Module):
    def __init__(self, in_channels, num_anchors, num_classes, num_convs=4, prior_probability=0.01, norm_layer=None):
        super().__init__()
        self.num_classes, self.num_anchors = num_classes, num_anchors
        self.conv = _gn_tower(in_channels, num_convs, norm_layer or partial(nn.GroupNorm, 32))
        self.cls_logits = nn.Conv2d(in_channels, num_anchors * num_classes, 3, 1, 1)
        nn.init.normal_(self.cls_logits.weight, std=0.01)
        nn.init.constant_(self.cls_logits.bias, -math.log((1 - prior_probability) / prior_probability))

    def forward_maps(self, x: List[Tensor]) -> List[Tensor]:
        """Per-level convolution outputs [B, A*C, H, W], before the permute + reshape of ``forward``."""
        return [self.cls_logits(self.conv(f)) for f in x]

    def forward(self, x: List[Tensor]) -> Tensor:
        return torch.cat([_to_anchor_major(m, self.num_classes) for m in self.forward_maps(x)], dim=1)


class FCOSRegressionHead(nn.Module):
    def __init__(self, in_channels, num_anchors, num_convs=4, norm_layer=None):
        super().__init__()
        self.conv = _gn_tower(in_channels, num_convs, norm_layer or partial(nn.GroupNorm, 32))
        self.bbox_reg = nn.Conv2d(in_channels, num_anchors * 4, 3, 1, 1)
        self.bbox_ctrness = nn.Conv2d(in_channels, num_anchors * 1, 3, 1, 1)
        for layer in (self.bbox_reg, self.bbox_ctrness):
            nn.init.normal_(layer.weight, std=0.01)
            nn.init.zeros_(layer.bias)

    def forward_maps(self, x: List[Tensor]) -> Tuple[List[Tensor], List[Tensor]]:
        """Per-level (regression [B, A*4, H, W], centre-ness [B, A, H, W]) before the permute + reshape."""
        regs, ctrs = [], []
        for f in x:
            t = self.conv(f)
            regs.append(nn.functional.relu(self.bbox_reg(t)))
            ctrs.append(self.bbox_ctrness(t))
        return regs, ctrs

    def forward(self, x: List[Tensor]) -> Tuple[Tensor, Tensor]:
        regs, ctrs = self.forward_maps(x)
        return (torch.cat([_to_anchor_major(r, 4) for r in regs], dim=1),
                torch.cat([_to_anchor_major(c, 1) for c in ctrs], dim=1))


class FCOSHead(nn.Module):
    def __init__(self, in_channels, num_anchors, num_classes, num_convs: Optional[int] = 4):
        super().__init__()
        self.box_coder = det_utils.BoxLinearCoder(normalize_by_size=True)
        self.classification_head = FCOSClassificationHead(in_channels, num_anchors, num_classes, num_convs)
        self.regression_head = FCOSRegressionHead(in_channels, num_anchors, num_convs)

    def compute_loss(self, targets, head_outputs, anchors, matched_idxs) -> Dict[str, Tensor]:
        cls_logits = head_outputs["cls_logits"]
        bbox_regression = head_outputs["bbox_regression"]
        bbox_ctrness = head_outputs["bbox_ctrness"]
        return det_utils.fcos_head_losses(cls_logits, bbox_regression, bbox_ctrness, list(anchors), list(matched_idxs),
                                          targets)

    def forward(self, x: List[Tensor]) -> Dict[str, Tensor]:
        cls_logits = self.classification_head(x)
        bbox_regression, bbox_ctrness = self.regression_head(x)
        return {"cls_logits": cls_logits, "bbox_regression": bbox_regression, "bbox_ctrness": bbox_ctrness}

    def forward_maps(self, x: List[Tensor]) -> Dict[str, List[Tensor]]:
        """Head outputs per level in the convolutions' own channel-major layout (inference path)."""
        regs, ctrs = self.regression_head.forward_maps(x)
        return {"cls_logits": self.classification_head.forward_maps(x), "bbox_regression": regs, "bbox_ctrness": ctrs}


class FCOS(nn.Module):
    def __init__(self, backbone, num_classes, min_size=800, max_size=1333, image_mean=None, image_std=None,
                 anchor_generator: Optional[AnchorGenerator] = None, head: Optional[nn.Module] = None,
                 center_sampling_radius: float = 1.5, score_thresh: float = 0.2, nms_thresh: float = 0.6,
                 detections_per_img: int = 100, topk_candidates: int = 1000, **kwargs):
        super().__init__()
        if not hasattr(backbone, "out_channels"):
            raise ValueError("backbone should contain an attribute out_channels specifying the number of "
                             "output channels (assumed to be the same for all the levels)")
        self.backbone = backbone
        if not isinstance(anchor_generator, (AnchorGenerator, type(None))):
            raise TypeError(f"anchor_generator should be of type AnchorGenerator or None, instead  got "
                            f"{type(anchor_generator)}")
        if anchor_generator is None:
            anchor_generator = AnchorGenerator(((8,), (16,), (32,), (64,), (128,)), ((1.0,),) * 5)
        self.anchor_generator = anchor_generator
        if self.anchor_generator.num_anchors_per_location()[0] != 1:
            raise ValueError(f"anchor_generator.num_anchors_per_location()[0] should be 1 instead of "
                             f"{anchor_generator.num_anchors_per_location()[0]}")
        if head is None:
            head = FCOSHead(backbone.out_channels, anchor_generator.num_anchors_per_location()[0], num_classes)
        self.head = head
        self.box_coder = det_utils.BoxLinearCoder(normalize_by_size=True)
        if image_mean is None:
            image_mean = [0.485, 0.456, 0.406]
        if image_std is None:
            image_std = [0.229, 0.224, 0.225]
        self.transform = GeneralizedRCNNTransform(min_size, max_size, image_mean, image_std, **kwargs)
        self.center_sampling_radius = center_sampling_radius
        self.score_thresh = score_thresh
        self.nms_thresh = nms_thresh
        self.detections_per_img = detections_per_img
        self.topk_candidates = topk_candidates

    def match_anchors(self, targets, anchors: List[Tensor], num_anchors_per_level: List[int]) -> List[Tensor]:
        gts = [t["boxes"] for t in targets]
        same = all(a.data_ptr() == anchors[0].data_ptr() and a.shape == anchors[0].shape for a in anchors)
        anc = anchors[0] if same else torch.stack(list(anchors))
        matched = kernels.match_fcos(anc, gts, num_anchors_per_level[0], num_anchors_per_level[-1],
                                     self.center_sampling_radius)
        return list(matched.unbind(0))

    def compute_loss(self, targets, head_outputs, anchors, num_anchors_per_level) -> Dict[str, Tensor]:
        matched_idxs = self.match_anchors(targets, anchors, num_anchors_per_level)
        return self.head.compute_loss(targets, head_outputs, anchors, matched_idxs)

    def postprocess_detections(self, head_outputs: Dict[str, List[Tensor]], anchors: List[List[Tensor]],
                               image_shapes: List[Tuple[int, int]]) -> List[Dict[str, Tensor]]:
        return postprocess_batched(head_outputs["cls_logits"], head_outputs["bbox_regression"], anchors,
                                   image_shapes, self.score_thresh, self.topk_candidates, self.nms_thresh,
                                   self.detections_per_img, _C.CODER_LINEAR, 0.0,
                                   ctr_levels=head_outputs["bbox_ctrness"])

    def forward(self, images: List[Tensor], targets: Optional[List[Dict[str, Tensor]]] = None):
        if self.training:
            if targets is None:
                torch._assert(False, "targets should not be none when in training mode")
            else:
                for target in targets:
                    boxes = target["boxes"]
                    torch._assert(isinstance(boxes, torch.Tensor), "Expected target boxes to be of type Tensor.")
                    torch._assert(len(boxes.shape) == 2 and boxes.shape[-1] == 4,
                                  f"Expected target boxes to be a tensor of shape [N, 4], got {boxes.shape}.")
        original_image_sizes = [(int(img.shape[-2]), int(img.shape[-1])) for img in images]
        images, targets = self.transform(images, targets)
        if targets is not None:
            for target_idx, target in enumerate(targets):
                boxes = target["boxes"]
                degenerate = boxes[:, 2:] <= boxes[:, :2]
                if degenerate.any():
                    bb_idx = torch.where(degenerate.any(dim=1))[0][0]
                    torch._assert(False, f"All bounding boxes should have positive height and width. Found invalid "
                                  f"box {boxes[bb_idx].tolist()} for target at index {target_idx}.")
        features = self.backbone(images.tensors)
        if isinstance(features, torch.Tensor):
            features = OrderedDict([("0", features)])
        features = list(features.values())
        if (not self.training and features[0].is_cuda and hasattr(self.head, "forward_maps")
                and type(self).postprocess_detections is FCOS.postprocess_detections
                and type(self.transform) is GeneralizedRCNNTransform):
            # inference: the candidate selection reads the heads' convolution outputs where they are
            maps = self.head.forward_maps(features)
            anchors = self.anchor_generator(images, features)
            num_classes = self.head.classification_head.num_classes
            per_level = [m.shape[1] // num_classes * m.shape[2] * m.shape[3] for m in maps["cls_logits"]]
            split_anchors = [list(a.split(per_level)) for a in anchors]
            detections = postprocess_batched(maps["cls_logits"], maps["bbox_regression"], split_anchors,
                                             images.image_sizes, self.score_thresh, self.topk_candidates,
                                             self.nms_thresh, self.detections_per_img, _C.CODER_LINEAR, 0.0,
                                             ctr_levels=maps["bbox_ctrness"], planar_classes=num_classes,
                                             box_scale=self.transform.box_scales(images.image_sizes, original_image_sizes,
                                                                                 features[0].device))
            return detections
        head_outputs = self.head(features)
        anchors = self.anchor_generator(images, features)
        num_anchors_per_level = [x.size(2) * x.size(3) * self.anchor_generator.num_anchors_per_location()[0]
                                 for x in features]
        if self.training:
            if targets is None:
                torch._assert(False, "targets should not be none when in training mode")
            return self.compute_loss(targets, head_outputs, anchors, num_anchors_per_level)
        split_head_outputs = {k: list(v.split(num_anchors_per_level, dim=1)) for k, v in head_outputs.items()}
        split_anchors = [list(a.split(num_anchors_per_level)) for a in anchors]
        detections = self.postprocess_detections(split_head_outputs, split_anchors, images.image_sizes)
        return self.transform.postprocess(detections, images.image_sizes, original_image_sizes)
