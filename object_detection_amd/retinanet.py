"""RetinaNet with the constructor, attributes and method signatures of
``torchvision.models.detection.retinanet`` (0.16.2), which the reference instantiates at
``object_detection/retinanet/retinanet_model.py:47-54`` and calls at ``object_detection/train.py:62,84``
and ``object_detection/predict.py:61``.

The box pipeline of the model -- anchors, anchor/GT matching in ``compute_loss``, and the whole of
``postprocess_detections`` -- runs in the HIP kernels of ``libod_boxes.so``, batched over images
(no per-image / per-level Python loops, one device->host read per batch for the detection counts).
The convolutional head layers are ordinary PyTorch modules; the head losses downstream of matching are fused
forward / backward kernels.
"""
from __future__ import annotations

import math
import threading
import weakref
from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Tuple

import torch
from torch import Tensor, nn

from . import _C, boxes as box_ops, det_utils, kernels
from .anchor_utils import AnchorGenerator
from .transform import GeneralizedRCNNTransform


def _conv_tower(in_channels: int, norm_layer: Optional[Callable[..., nn.Module]]) -> nn.Sequential:
    """Four 3x3 conv (+norm) + ReLU blocks; block layout [conv, (norm), relu] keeps state-dict keys
    ``conv.<i>.0.weight`` / ``conv.<i>.1.weight`` like torchvision's Conv2dNormActivation."""
    blocks = []
    for _ in range(4):
        layers: List[nn.Module] = [nn.Conv2d(in_channels, in_channels, 3, 1, 1, bias=norm_layer is None)]
        if norm_layer is not None:
            layers.append(norm_layer(in_channels))
        layers.append(nn.ReLU(inplace=True))
        blocks.append(nn.Sequential(*layers))
    tower = nn.Sequential(*blocks)
    for m in tower.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.normal_(m.weight, std=0.01)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
    return tower


def _to_anchor_major(x: Tensor, per_anchor: int) -> Tensor:
    # (B, A*K, H, W) -> (B, H*W*A, K): memory order (h, w, a, k) equals the anchor order (y, x, a)
    B, _, H, W = x.shape
    return x.view(B, -1, per_anchor, H, W).permute(0, 3, 4, 1, 2).reshape(B, -1, per_anchor)


class RetinaNetClassificationHead(nn.Module):
    def __init__(self, in_channels, num_anchors, num_classes, prior_probability=0.01,
                 norm_layer: Optional[Callable[..., nn.Module]] = None):
        super().__init__()
        self.conv = _conv_tower(in_channels, norm_layer)
        self.cls_logits = nn.Conv2d(in_channels, num_anchors * num_classes, kernel_size=3, stride=1, padding=1)
        nn.init.normal_(self.cls_logits.weight, std=0.01)
        nn.init.constant_(self.cls_logits.bias, -math.log((1 - prior_probability) / prior_probability))
        self.num_classes = num_classes
        self.num_anchors = num_anchors
        self.BETWEEN_THRESHOLDS = det_utils.Matcher.BETWEEN_THRESHOLDS

    def compute_loss(self, targets, head_outputs, matched_idxs) -> Tensor:
        return det_utils.retinanet_classification_loss(head_outputs["cls_logits"], list(matched_idxs),
                                                       [t["labels"] for t in targets])

    def forward_maps(self, x: List[Tensor]) -> List[Tensor]:
        """Per-level convolution outputs [B, A*C, H, W], before the permute + reshape of ``forward``."""
        return [self.cls_logits(self.conv(f)) for f in x]

    def forward(self, x: List[Tensor]) -> Tensor:
        return torch.cat([_to_anchor_major(m, self.num_classes) for m in self.forward_maps(x)], dim=1)


class RetinaNetRegressionHead(nn.Module):
    def __init__(self, in_channels, num_anchors, norm_layer: Optional[Callable[..., nn.Module]] = None):
        super().__init__()
        self.conv = _conv_tower(in_channels, norm_layer)
        self.bbox_reg = nn.Conv2d(in_channels, num_anchors * 4, kernel_size=3, stride=1, padding=1)
        nn.init.normal_(self.bbox_reg.weight, std=0.01)
        nn.init.zeros_(self.bbox_reg.bias)
        self.box_coder = det_utils.BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))
        self._loss_type = "l1"

    def compute_loss(self, targets, head_outputs, anchors, matched_idxs) -> Tensor:
        if len(targets) == 0:
            return head_outputs["bbox_regression"].sum() * 0.0
        return det_utils.retinanet_regression_loss(self._loss_type, self.box_coder, head_outputs["bbox_regression"],
                                                   list(anchors), list(matched_idxs), [t["boxes"] for t in targets])

    def forward_maps(self, x: List[Tensor]) -> List[Tensor]:
        """Per-level convolution outputs [B, A*4, H, W], before the permute + reshape of ``forward``."""
        return [self.bbox_reg(self.conv(f)) for f in x]

    def forward(self, x: List[Tensor]) -> Tensor:
        return torch.cat([_to_anchor_major(m, 4) for m in self.forward_maps(x)], dim=1)


class RetinaNetHead(nn.Module):
    def __init__(self, in_channels, num_anchors, num_classes, norm_layer: Optional[Callable[..., nn.Module]] = None):
        super().__init__()
        self.classification_head = RetinaNetClassificationHead(in_channels, num_anchors, num_classes,
                                                               norm_layer=norm_layer)
        self.regression_head = RetinaNetRegressionHead(in_channels, num_anchors, norm_layer=norm_layer)

    def compute_loss(self, targets, head_outputs, anchors, matched_idxs) -> Dict[str, Tensor]:
        return {
            "classification": self.classification_head.compute_loss(targets, head_outputs, matched_idxs),
            "bbox_regression": self.regression_head.compute_loss(targets, head_outputs, anchors, matched_idxs),
        }

    def forward(self, x: List[Tensor]) -> Dict[str, Tensor]:
        return {"cls_logits": self.classification_head(x), "bbox_regression": self.regression_head(x)}

    def forward_maps(self, x: List[Tensor]) -> Dict[str, List[Tensor]]:
        """Head outputs per level in the convolutions' own channel-major layout (inference path: the candidate
        selection reads them in place instead of the permuted copies ``forward`` makes)."""
        return {"cls_logits": self.classification_head.forward_maps(x),
                "bbox_regression": self.regression_head.forward_maps(x)}


def _shared_levels(anchors: List[List[Tensor]]) -> Optional[List[Tensor]]:
    """Per-level anchors if every image uses the same tensors (what AnchorGenerator hands out)."""
    first = anchors[0]
    for other in anchors[1:]:
        if other is first:          # AnchorGenerator returns the same list object for every image
            continue
        for a, b in zip(first, other):
            if a.data_ptr() != b.data_ptr() or a.shape != b.shape:
                return None
    return list(first)


_hw_tables: Dict[tuple, Tensor] = {}


def _image_hw_tensor(image_shapes: List[Tuple[int, int]], device: torch.device) -> Tensor:
    """int32 [B, 2] (height, width) on the device; looked up by the shapes themselves (a training or prediction loop
    passes the same list of sizes step after step)."""
    key = (device.index, kernels._raw_stream(device), tuple(image_shapes))
    t = _hw_tables.get(key)
    if t is None:
        if len(_hw_tables) > 256:
            _hw_tables.clear()
        flat = [int(v) for hw in image_shapes for v in hw]
        t = _hw_tables[key] = kernels._device_ints(flat, device).view(-1, 2)
    return kernels._held(t)


_COUNT_ROWS = 64                    # pinned rows for detection counts on their way to the host
_count_pool: Dict[int, Tensor] = {}  # batch size -> pinned int32 [_COUNT_ROWS, batch]
_count_owner: Dict[tuple, "weakref.ref"] = {}
_count_next: Dict[int, int] = {}
_count_events: Dict[tuple, list] = {}  # (device, batch size) -> one event per pinned row
_count_lock = threading.Lock()         # the hand-out of pinned rows: two threads must not receive the same row


class _PendingBatch:
    """Padded NMS outputs of a batch and the detection counts on their way to the host (asynchronous copy into a row of
    pinned memory + an event recorded behind it).  ``counts()`` waits for that event once.  The pinned rows are handed
    out in turn; a row whose previous batch was never looked at is read out (one wait) before it is reused."""
    __slots__ = ("boxes", "scores", "labels", "counts_dev", "counts_host", "event", "_counts", "_rows", "__weakref__")

    def __init__(self, boxes: Tensor, scores: Tensor, labels: Tensor, counts_dev: Tensor):
        self.boxes, self.scores, self.labels, self.counts_dev = boxes, scores, labels, counts_dev
        B = int(counts_dev.shape[0])
        with _count_lock:
            pool = _count_pool.get(B)
            if pool is None:
                pool = _count_pool[B] = torch.empty((_COUNT_ROWS, B), dtype=torch.int32).pin_memory()
            row = _count_next.get(B, 0)
            _count_next[B] = (row + 1) % _COUNT_ROWS
            previous = _count_owner.get((B, row))
            _count_owner[(B, row)] = weakref.ref(self)
        previous = previous() if previous is not None else None
        if previous is not None:
            previous.counts()
        self.counts_host = pool[row]
        self.counts_host.copy_(counts_dev, non_blocking=True)
        # one event per pinned row, recorded again at every reuse (the row's previous batch has been read out above)
        events = _count_events.setdefault((counts_dev.device.index, B), [None] * _COUNT_ROWS)
        if events[row] is None:
            events[row] = torch.cuda.Event()
        self.event = events[row]
        self.event.record()
        self._counts = None
        self._rows = None

    def counts(self) -> List[int]:
        if self._counts is None:
            self.event.synchronize()
            self._counts = self.counts_host.tolist()
        return self._counts

    def rows(self):
        """Per-image (boxes [d, 4], scores [d], labels [d]) views cut to size, made for the whole batch at the first
        request: one split of each flattened padded tensor into (kept, padding) pieces per image -- three calls into
        torch instead of three slices per image."""
        if self._rows is None:
            counts = self.counts()
            B, D = int(self.scores.shape[0]), int(self.scores.shape[1])
            short = [i for i, c in enumerate(counts) if c != D]
            if 4 * len(short) <= B:
                # (nearly) every image fills its detections_per_img rows: the rows of the padded tensors are the
                # results as they are -- one unbind per tensor, and a slice for the few images that stay below the cap
                rb, rs, rl = list(self.boxes.unbind(0)), list(self.scores.unbind(0)), list(self.labels.unbind(0))
                for i in short:
                    c = counts[i]
                    rb[i], rs[i], rl[i] = rb[i][:c], rs[i][:c], rl[i][:c]
                self._rows = (rb, rs, rl)
                return self._rows
            sizes = [0] * (2 * B)
            sizes[0::2] = counts
            sizes[1::2] = [D - c for c in counts]
            self._rows = (self.boxes.reshape(B * D, 4).split_with_sizes(sizes)[0::2],
                          self.scores.reshape(B * D).split_with_sizes(sizes)[0::2],
                          self.labels.reshape(B * D).split_with_sizes(sizes)[0::2])
        return self._rows


class _ReadyDetections(dict):
    """What a ``Detections`` object becomes once its tensors are cut out: a plain dictionary again (no Python-level
    method between the caller and ``dict``), which keeps answering ``defer_box_scale`` with False."""

    __slots__ = ("_pending", "_index", "_scale")             # the layout of Detections: the class is swapped in place
    __hash__ = None

    def defer_box_scale(self, scale: Tensor) -> bool:
        return False


class Detections(dict):
    """``{"boxes": [d, 4], "scores": [d], "labels": [d]}`` of one image, as ``postprocess_detections`` returns it.  The
    number of detections d is only known on the GPU when the call returns; the three tensors are cut out of the padded
    outputs of the batch at the first access to the dictionary, which waits for the counts of the batch (one event;
    the first access to any image of the batch also cuts the three padded tensors for all its images, three calls into
    torch) -- the call itself never makes the host wait for the GPU.  A filled dictionary turns into a plain ``dict``
    subclass without overridden accessors."""

    __slots__ = ("_pending", "_index", "_scale")

    def __init__(self, pending: _PendingBatch, index: int):
        self._pending = pending
        self._index = index
        self._scale = None

    def defer_box_scale(self, scale: Tensor) -> bool:
        """Multiply the boxes by ``scale`` [4] when they are cut out (the rescale of ``transform.postprocess``);
        False if the dictionary is already materialised."""
        if self._pending is None:
            return False
        self._scale = scale if self._scale is None else self._scale * scale
        return True

    def _fill(self, boxes: Tensor, scores: Tensor, labels: Tensor) -> None:
        if self._pending is None:
            return
        self._pending = None
        if self._scale is not None:
            boxes = boxes * self._scale
        dict.update(self, boxes=boxes, scores=scores, labels=labels)
        self.__class__ = _ReadyDetections

    def _materialize(self) -> None:
        p = self._pending
        if p is None:
            return
        rb, rs, rl = p.rows()
        i = self._index
        self._fill(rb[i], rs[i], rl[i])

    def __getitem__(self, key):
        self._materialize()
        return dict.__getitem__(self, key)

    def __setitem__(self, key, value):
        self._materialize()
        dict.__setitem__(self, key, value)

    def __iter__(self):
        self._materialize()
        return dict.__iter__(self)

    def __len__(self):
        self._materialize()
        return dict.__len__(self)

    def __contains__(self, key):
        self._materialize()
        return dict.__contains__(self, key)

    def __repr__(self):
        self._materialize()
        return dict.__repr__(self)

    def __eq__(self, other):
        self._materialize()
        return dict.__eq__(self, other)

    __hash__ = None

    def keys(self):
        self._materialize()
        return dict.keys(self)

    def values(self):
        self._materialize()
        return dict.values(self)

    def items(self):
        self._materialize()
        return dict.items(self)

    def get(self, key, default=None):
        self._materialize()
        return dict.get(self, key, default)

    def pop(self, *args):
        self._materialize()
        return dict.pop(self, *args)

    def update(self, *args, **kwargs):
        self._materialize()
        dict.update(self, *args, **kwargs)

    def copy(self):
        self._materialize()
        return dict.copy(self)

    def __ne__(self, other):
        self._materialize()
        return dict.__ne__(self, other)

    def __delitem__(self, key):
        self._materialize()
        dict.__delitem__(self, key)

    def __reversed__(self):
        self._materialize()
        return dict.__reversed__(self)

    def __or__(self, other):
        self._materialize()
        return dict.__or__(self, other)

    def __ror__(self, other):
        self._materialize()
        return dict.__ror__(self, other)

    def __ior__(self, other):
        self._materialize()
        dict.update(self, other)
        return self

    def setdefault(self, key, default=None):
        self._materialize()
        return dict.setdefault(self, key, default)

    def popitem(self):
        self._materialize()
        return dict.popitem(self)

    def clear(self):
        self._materialize()
        dict.clear(self)


def postprocess_batched(cls_levels, reg_levels, anchors, image_shapes, score_thresh, topk_candidates, nms_thresh,
                        detections_per_img, coder, bbox_xform_clip, ctr_levels=None,
                        nms_mode=_C.NMS_AUTO_GPU | _C.NMS_THRESHOLD_FLOAT,
                        return_padded: bool = False, planar_classes: Optional[int] = None,
                        box_scale: Optional[Tensor] = None):
    """Shared tail of RetinaNet / FCOS postprocess_detections: two library calls for the whole batch.
    With ``planar_classes`` (= number of classes) the level tensors are the heads' convolution outputs
    [B, A*K, H, W] instead of the permuted [B, H*W*A, K].  ``box_scale`` [B, 4] multiplies the kept boxes of each image
    (the rescale of ``transform.postprocess``).  ``nms_mode``: the default is what torchvision does for tensors on a
    GPU -- per-class branch above 20000 box coordinates, fp32 comparison with float(nms_thresh) as in its CUDA
    kernel; ``_C.NMS_AUTO_CPU`` is its behaviour on the CPU (4000, comparison in double).
    Returns the list of per-image ``Detections`` (cut to size on first access, no host wait here), or with
    ``return_padded`` the padded tensors (boxes [B, D, 4], scores [B, D], labels [B, D], counts [B])."""
    shared = _shared_levels(anchors)
    if shared is None:
        shared = [torch.stack([a[l] for a in anchors]) for l in range(len(anchors[0]))]
    dev = cls_levels[0].device
    hw = _image_hw_tensor(image_shapes, dev)
    K = int(topk_candidates)
    if planar_classes is not None:
        cb, cs, cl, _, cc = kernels.select_candidates_planar(cls_levels, reg_levels, shared, hw, planar_classes,
                                                             score_thresh, K, coder, bbox_xform_clip, ctr_maps=ctr_levels,
                                                             reuse_outputs=True)
    else:
        cb, cs, cl, _, cc = kernels.select_candidates(cls_levels, reg_levels, shared, hw, score_thresh, K, coder,
                                                      bbox_xform_clip, ctr_levels=ctr_levels, reuse_outputs=True)
    _, ob, os_, ol, oc = kernels.batched_nms_padded(cb, cs, cl, cc, K, True, nms_thresh, nms_mode,
                                                    int(detections_per_img), want_keep_idx=False)
    if box_scale is not None:
        ob = ob * box_scale[:, None, :]
    if return_padded:
        return ob, os_, ol, oc
    pending = _PendingBatch(ob, os_, ol, oc)
    return [Detections(pending, i) for i in range(ob.shape[0])]


class RetinaNet(nn.Module):
    def __init__(self, backbone, num_classes, min_size=800, max_size=1333, image_mean=None, image_std=None,
                 anchor_generator=None, head=None, proposal_matcher=None, score_thresh=0.05, nms_thresh=0.5,
                 detections_per_img=300, fg_iou_thresh=0.5, bg_iou_thresh=0.4, topk_candidates=1000, **kwargs):
        super().__init__()
        if not hasattr(backbone, "out_channels"):
            raise ValueError("backbone should contain an attribute out_channels specifying the number of "
                             "output channels (assumed to be the same for all the levels)")
        self.backbone = backbone
        if not isinstance(anchor_generator, (AnchorGenerator, type(None))):
            raise TypeError(f"anchor_generator should be of type AnchorGenerator or None instead of "
                            f"{type(anchor_generator)}")
        if anchor_generator is None:
            sizes = tuple((x, int(x * 2 ** (1.0 / 3)), int(x * 2 ** (2.0 / 3))) for x in [32, 64, 128, 256, 512])
            anchor_generator = AnchorGenerator(sizes, ((0.5, 1.0, 2.0),) * len(sizes))
        self.anchor_generator = anchor_generator
        if head is None:
            head = RetinaNetHead(backbone.out_channels, anchor_generator.num_anchors_per_location()[0], num_classes)
        self.head = head
        if proposal_matcher is None:
            proposal_matcher = det_utils.Matcher(fg_iou_thresh, bg_iou_thresh, allow_low_quality_matches=True)
        self.proposal_matcher = proposal_matcher
        self.box_coder = det_utils.BoxCoder(weights=(1.0, 1.0, 1.0, 1.0))
        if image_mean is None:
            image_mean = [0.485, 0.456, 0.406]
        if image_std is None:
            image_std = [0.229, 0.224, 0.225]
        self.transform = GeneralizedRCNNTransform(min_size, max_size, image_mean, image_std, **kwargs)
        self.score_thresh = score_thresh
        self.nms_thresh = nms_thresh
        self.detections_per_img = detections_per_img
        self.topk_candidates = topk_candidates
        self._has_warned = False

    # ---- training side: anchor / ground-truth matching ------------------------------------------------
    def match_anchors(self, targets: List[Dict[str, Tensor]], anchors: List[Tensor]) -> List[Tensor]:
        """matched_idxs per image: one fused box_iou+Matcher launch pair for the whole batch."""
        pm = self.proposal_matcher
        gts = [t["boxes"] for t in targets]
        if isinstance(pm, det_utils.Matcher):
            first = anchors[0]
            same = all(a is first or (a.data_ptr() == first.data_ptr() and a.shape == first.shape) for a in anchors)
            anc = anchors[0] if same else torch.stack(list(anchors))
            matched = kernels.match_retinanet(anc, gts, pm.high_threshold, pm.low_threshold,
                                              pm.allow_low_quality_matches)
            return list(matched.unbind(0))
        out = []   # user-supplied matcher object: matrix form, image by image
        for anchors_per_image, gt in zip(anchors, gts):
            if gt.numel() == 0:
                out.append(torch.full((anchors_per_image.size(0),), -1, dtype=torch.int64,
                                      device=anchors_per_image.device))
                continue
            out.append(pm(box_ops.box_iou(gt, anchors_per_image)))
        return out

    def compute_loss(self, targets, head_outputs, anchors) -> Dict[str, Tensor]:
        matched_idxs = self.match_anchors(targets, anchors)
        return self.head.compute_loss(targets, head_outputs, anchors, matched_idxs)

    # ---- inference side ------------------------------------------------------------------------------------
    def postprocess_detections(self, head_outputs: Dict[str, List[Tensor]], anchors: List[List[Tensor]],
                               image_shapes: List[Tuple[int, int]]) -> List[Dict[str, Tensor]]:
        return postprocess_batched(head_outputs["cls_logits"], head_outputs["bbox_regression"], anchors,
                                   image_shapes, self.score_thresh, self.topk_candidates, self.nms_thresh,
                                   self.detections_per_img, _C.CODER_BOXCODER, self.box_coder.bbox_xform_clip)

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
        original_image_sizes: List[Tuple[int, int]] = []
        for img in images:
            val = img.shape[-2:]
            torch._assert(len(val) == 2, f"expecting the last two dimensions of the Tensor to be H and W instead got {img.shape[-2:]}")
            original_image_sizes.append((int(val[0]), int(val[1])))
        images, targets = self.transform(images, targets)
        if targets is not None:
            for target_idx, target in enumerate(targets):
                boxes = target["boxes"]
                degenerate = boxes[:, 2:] <= boxes[:, :2]
                if degenerate.any():
                    bb_idx = torch.where(degenerate.any(dim=1))[0][0]
                    degen_bb: List[float] = boxes[bb_idx].tolist()
                    torch._assert(False, "All bounding boxes should have positive height and width."
                                  f" Found invalid box {degen_bb} for target at index {target_idx}.")
        features = self.backbone(images.tensors)
        if isinstance(features, torch.Tensor):
            features = OrderedDict([("0", features)])
        features = list(features.values())
        if (not self.training and features[0].is_cuda and hasattr(self.head, "forward_maps")
                and type(self).postprocess_detections is RetinaNet.postprocess_detections
                and type(self.transform) is GeneralizedRCNNTransform):
            # inference: the candidate selection reads the heads' convolution outputs where they are
            maps = self.head.forward_maps(features)
            anchors = self.anchor_generator(images, features)
            num_classes = self.head.classification_head.num_classes
            per_level = [m.shape[1] // num_classes * m.shape[2] * m.shape[3] for m in maps["cls_logits"]]
            split_anchors = [list(a.split(per_level)) for a in anchors]
            detections = postprocess_batched(maps["cls_logits"], maps["bbox_regression"], split_anchors,
                                             images.image_sizes, self.score_thresh, self.topk_candidates,
                                             self.nms_thresh, self.detections_per_img, _C.CODER_BOXCODER,
                                             self.box_coder.bbox_xform_clip, planar_classes=num_classes,
                                             box_scale=self.transform.box_scales(images.image_sizes, original_image_sizes,
                                                                                 features[0].device))
            return detections
        head_outputs = self.head(features)
        anchors = self.anchor_generator(images, features)
        losses: Dict[str, Tensor] = {}
        detections: List[Dict[str, Tensor]] = []
        if self.training:
            if targets is None:
                torch._assert(False, "targets should not be none when in training mode")
            else:
                losses = self.compute_loss(targets, head_outputs, anchors)
            return losses
        num_anchors_per_level = [x.size(2) * x.size(3) for x in features]
        HW = 0
        for v in num_anchors_per_level:
            HW += v
        HWA = head_outputs["cls_logits"].size(1)
        A = HWA // HW
        num_anchors_per_level = [hw * A for hw in num_anchors_per_level]
        split_head_outputs: Dict[str, List[Tensor]] = {}
        for k in head_outputs:
            split_head_outputs[k] = list(head_outputs[k].split(num_anchors_per_level, dim=1))
        split_anchors = [list(a.split(num_anchors_per_level)) for a in anchors]
        detections = self.postprocess_detections(split_head_outputs, split_anchors, images.image_sizes)
        return self.transform.postprocess(detections, images.image_sizes, original_image_sizes)
