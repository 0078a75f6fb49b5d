"""CPU restatement of the torchvision 0.16.2 detection-head box pipeline (TEST INFRASTRUCTURE).

PARITY UNPINNED: the reference (Kotaro-Kuroda/object_detection v0.3.5) keeps none of this
arithmetic in its own tree -- it constructs ``torchvision==0.16.2`` ``RetinaNet`` / ``FCOS``
objects (reference ``object_detection/retinanet/retinanet_model.py:4-7,25-30,44-54,57,60``) and
calls them (``object_detection/train.py:62,84``, ``object_detection/predict.py:61,72``).
torchvision is not installed in the build container, not vendored by the reference, and the
reference has no tests or golden vectors.  Every function below therefore restates the
*published* torchvision 0.16.2 algorithm from its public source, module by module, and is
anchored only by hand-derived known-answer tests (tests/test_oracle_kat.py).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product (``object_detection_amd``) never does.

Two math modes for the two transcendental calls on the path (``torch.sigmoid`` in the score
filter, ``torch.exp`` in ``BoxCoder.decode_single``):
  * ``math="torch"`` -- literal restatement (``torch.sigmoid`` / ``torch.exp``).
  * ``math="spec"``  -- the op-by-op fp32 algorithm of ``oracle/spec_math.py`` that the HIP
    kernels also implement, so that candidate order, decoded boxes and hence NMS keep indices
    are bit-reproducible.  The two modes agree to ~2e-7 relative.

Tie rule adopted where torch leaves order unspecified (``Tensor.topk`` among equal scores,
non-stable ``sort`` in ``_batched_nms_vanilla``): descending score, then ascending index.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import spec_math

BELOW_LOW_THRESHOLD = -1   # torchvision/models/detection/_utils.py: Matcher.BELOW_LOW_THRESHOLD
BETWEEN_THRESHOLDS = -2    # torchvision/models/detection/_utils.py: Matcher.BETWEEN_THRESHOLDS


# --------------------------------------------------------------------------------------------
# math-mode helpers
# --------------------------------------------------------------------------------------------
def _exp(x: Tensor, mode: str) -> Tensor:
    if mode == "torch":
        return torch.exp(x)
    return torch.from_numpy(spec_math.od_expf(x.detach().cpu().numpy())).reshape(x.shape)


def _sigmoid(x: Tensor, mode: str) -> Tensor:
    if mode == "torch":
        return torch.sigmoid(x)
    return torch.from_numpy(spec_math.od_sigmoidf(x.detach().cpu().numpy())).reshape(x.shape)


def _sqrt(x: Tensor, mode: str) -> Tensor:
    # torch's CPU sqrt kernel is not correctly rounded on every build (it differs from IEEE sqrtf in
    # ~0.6 % of fp32 inputs here); the specified pipeline uses the IEEE correctly rounded square root.
    if mode == "torch":
        return torch.sqrt(x)
    return torch.from_numpy(np.sqrt(x.detach().cpu().numpy().astype(np.float32))).reshape(x.shape)


# --------------------------------------------------------------------------------------------
# torchvision/ops/boxes.py
# --------------------------------------------------------------------------------------------
def _upcast(t: Tensor) -> Tensor:
    # torchvision/ops/_utils.py:_upcast -- protects from overflow: fp16 -> fp32, small ints -> int
    if t.is_floating_point():
        return t if t.dtype in (torch.float32, torch.float64) else t.float()
    return t if t.dtype in (torch.int32, torch.int64) else t.int()


def box_area(boxes: Tensor) -> Tensor:
    boxes = _upcast(boxes)
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def _box_inter_union(boxes1: Tensor, boxes2: Tensor) -> Tuple[Tensor, Tensor]:
    area1 = box_area(boxes1)
    area2 = box_area(boxes2)
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])  # [N,M,2]
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])  # [N,M,2]
    wh = _upcast(rb - lt).clamp(min=0)  # [N,M,2]
    inter = wh[:, :, 0] * wh[:, :, 1]  # [N,M]
    union = area1[:, None] + area2 - inter
    return inter, union


def box_iou(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    inter, union = _box_inter_union(boxes1, boxes2)
    return inter / union


def clip_boxes_to_image(boxes: Tensor, size: Tuple[int, int]) -> Tensor:
    dim = boxes.dim()
    boxes_x = boxes[..., 0::2]
    boxes_y = boxes[..., 1::2]
    height, width = size
    boxes_x = boxes_x.clamp(min=0, max=width)
    boxes_y = boxes_y.clamp(min=0, max=height)
    clipped = torch.stack((boxes_x, boxes_y), dim=dim)
    return clipped.reshape(boxes.shape)


# Order among EQUAL scores, where torch leaves it open (``Tensor.topk``, the non-stable ``sort`` calls of nms and
# _batched_nms_vanilla).  "ascending" (index) is the adopted rule, which the HIP kernels implement.  "descending" exists
# to measure how much of a result hangs on that choice: ``with tie_order("descending"): ...`` reruns the restatement
# with every such tie broken the other way round.
_TIE_ORDER = "ascending"


class tie_order:
    def __init__(self, order: str):
        assert order in ("ascending", "descending")
        self.order = order

    def __enter__(self):
        global _TIE_ORDER
        self.previous, _TIE_ORDER = _TIE_ORDER, self.order
        return self

    def __exit__(self, *exc):
        global _TIE_ORDER
        _TIE_ORDER = self.previous
        return False


def _argsort_desc(scores: Tensor) -> Tensor:
    """Indices by descending score; equal scores by ascending index (or descending, see tie_order)."""
    if _TIE_ORDER == "ascending":
        return torch.sort(scores, dim=0, descending=True, stable=True)[1]
    n = scores.shape[0]
    return (n - 1) - torch.sort(scores.flip(0), dim=0, descending=True, stable=True)[1]


def nms(boxes: Tensor, scores: Tensor, iou_threshold: float, float_rule: bool = False) -> Tensor:
    """torchvision/csrc/ops/cpu/nms_kernel.cpp:nms_kernel_impl, vectorised over the inner loop.

    Order = stable descending sort of scores; a later box j is suppressed by a kept box i iff
    fp32 ``inter / (area_i + area_j - inter)`` promoted to double is ``>`` the double threshold.
    ``float_rule``: the decision of torchvision/csrc/ops/cuda/nms_kernel.cu instead -- the fp32 IoU is compared with
    the threshold rounded to float (same order, same IoU formula); the two differ when IoU == float(threshold).
    """
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    b = boxes.detach().cpu().float().numpy()
    s = scores.detach().cpu().float()
    x1, y1, x2, y2 = (np.ascontiguousarray(b[:, i]) for i in range(4))
    areas = (x2 - x1) * (y2 - y1)
    order = _argsort_desc(s).numpy()
    n = b.shape[0]
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    zero = np.float32(0)
    thr = float(iou_threshold)
    with np.errstate(invalid="ignore", divide="ignore"):
        for _i in range(n):
            i = order[_i]
            if suppressed[i]:
                continue
            keep.append(i)
            rest = order[_i + 1:]
            xx1 = np.maximum(x1[i], x1[rest])
            yy1 = np.maximum(y1[i], y1[rest])
            xx2 = np.minimum(x2[i], x2[rest])
            yy2 = np.minimum(y2[i], y2[rest])
            w = np.maximum(zero, xx2 - xx1)
            h = np.maximum(zero, yy2 - yy1)
            inter = w * h
            ovr = inter / (areas[i] + areas[rest] - inter)
            if float_rule:
                suppressed[rest[ovr > np.float32(thr)]] = True
            else:
                suppressed[rest[ovr.astype(np.float64) > thr]] = True
    return torch.as_tensor(np.asarray(keep, dtype=np.int64))


def _batched_nms_coordinate_trick(boxes, scores, idxs, iou_threshold, float_rule: bool = False):
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    boxes_for_nms = boxes + offsets[:, None]
    return nms(boxes_for_nms, scores, iou_threshold, float_rule)


def _batched_nms_vanilla(boxes, scores, idxs, iou_threshold, float_rule: bool = False):
    keep_mask = torch.zeros_like(scores, dtype=torch.bool)
    for class_id in torch.unique(idxs):
        curr_indices = torch.where(idxs == class_id)[0]
        curr_keep = nms(boxes[curr_indices], scores[curr_indices], iou_threshold, float_rule)
        keep_mask[curr_indices[curr_keep]] = True
    keep_indices = torch.where(keep_mask)[0]
    # torchvision calls a non-stable sort here; adopted tie rule: ascending index.
    return keep_indices[_argsort_desc(scores[keep_indices])]


def batched_nms(boxes, scores, idxs, iou_threshold, device_rule: str = "cpu"):
    """torchvision/ops/boxes.py:batched_nms; ``device_rule`` plays ``boxes.device.type``: it selects the branch
    (numel > 4000 on "cpu", > 20000 otherwise) and the native kernel whose threshold rule applies ("cpu": double
    comparison of nms_kernel.cpp; "gpu": float comparison of nms_kernel.cu; "gpu-cpu-threshold": the gpu branch rule
    with the cpu kernel's comparison)."""
    limit = 4000 if device_rule == "cpu" else 20000
    float_rule = device_rule == "gpu"
    if boxes.numel() > limit:
        return _batched_nms_vanilla(boxes, scores, idxs, iou_threshold, float_rule)
    return _batched_nms_coordinate_trick(boxes, scores, idxs, iou_threshold, float_rule)


# --------------------------------------------------------------------------------------------
# torchvision/models/detection/_utils.py
# --------------------------------------------------------------------------------------------
class Matcher:
    def __init__(self, high_threshold: float, low_threshold: float, allow_low_quality_matches: bool = False):
        assert low_threshold <= high_threshold
        self.high_threshold = high_threshold
        self.low_threshold = low_threshold
        self.allow_low_quality_matches = allow_low_quality_matches

    def __call__(self, match_quality_matrix: Tensor) -> Tensor:
        if match_quality_matrix.numel() == 0:
            if match_quality_matrix.shape[0] == 0:
                raise ValueError("No ground-truth boxes available for one of the images during training")
            raise ValueError("No proposal boxes available for one of the images during training")
        matched_vals, matches = match_quality_matrix.max(dim=0)
        all_matches = matches.clone() if self.allow_low_quality_matches else None
        below = matched_vals < self.low_threshold
        between = (matched_vals >= self.low_threshold) & (matched_vals < self.high_threshold)
        matches[below] = BELOW_LOW_THRESHOLD
        matches[between] = BETWEEN_THRESHOLDS
        if self.allow_low_quality_matches:
            highest_quality_foreach_gt, _ = match_quality_matrix.max(dim=1)
            pairs = torch.where(match_quality_matrix == highest_quality_foreach_gt[:, None])
            pred_inds_to_update = pairs[1]
            matches[pred_inds_to_update] = all_matches[pred_inds_to_update]
        return matches


class BoxCoder:
    def __init__(self, weights=(1.0, 1.0, 1.0, 1.0), bbox_xform_clip: float = math.log(1000.0 / 16)):
        self.weights = weights
        self.bbox_xform_clip = bbox_xform_clip

    def encode_single(self, reference_boxes: Tensor, proposals: Tensor) -> Tensor:
        """torchvision _utils.encode_boxes: regression targets of ``reference_boxes`` relative to ``proposals``."""
        wx, wy, ww, wh = self.weights
        ex_widths = proposals[:, 2] - proposals[:, 0]
        ex_heights = proposals[:, 3] - proposals[:, 1]
        ex_ctr_x = proposals[:, 0] + 0.5 * ex_widths
        ex_ctr_y = proposals[:, 1] + 0.5 * ex_heights
        gt_widths = reference_boxes[:, 2] - reference_boxes[:, 0]
        gt_heights = reference_boxes[:, 3] - reference_boxes[:, 1]
        gt_ctr_x = reference_boxes[:, 0] + 0.5 * gt_widths
        gt_ctr_y = reference_boxes[:, 1] + 0.5 * gt_heights
        return torch.stack((wx * (gt_ctr_x - ex_ctr_x) / ex_widths, wy * (gt_ctr_y - ex_ctr_y) / ex_heights,
                            ww * torch.log(gt_widths / ex_widths), wh * torch.log(gt_heights / ex_heights)), dim=1)

    def decode_single(self, rel_codes: Tensor, boxes: Tensor, math: str = "torch") -> Tensor:
        boxes = boxes.to(rel_codes.dtype)
        widths = boxes[:, 2] - boxes[:, 0]
        heights = boxes[:, 3] - boxes[:, 1]
        ctr_x = boxes[:, 0] + 0.5 * widths
        ctr_y = boxes[:, 1] + 0.5 * heights
        wx, wy, ww, wh = self.weights
        dx = rel_codes[:, 0::4] / wx
        dy = rel_codes[:, 1::4] / wy
        dw = rel_codes[:, 2::4] / ww
        dh = rel_codes[:, 3::4] / wh
        dw = torch.clamp(dw, max=self.bbox_xform_clip)
        dh = torch.clamp(dh, max=self.bbox_xform_clip)
        pred_ctr_x = dx * widths[:, None] + ctr_x[:, None]
        pred_ctr_y = dy * heights[:, None] + ctr_y[:, None]
        pred_w = _exp(dw, math) * widths[:, None]
        pred_h = _exp(dh, math) * heights[:, None]
        c_to_c_h = torch.tensor(0.5, dtype=pred_ctr_y.dtype) * pred_h
        c_to_c_w = torch.tensor(0.5, dtype=pred_ctr_x.dtype) * pred_w
        pred_boxes1 = pred_ctr_x - c_to_c_w
        pred_boxes2 = pred_ctr_y - c_to_c_h
        pred_boxes3 = pred_ctr_x + c_to_c_w
        pred_boxes4 = pred_ctr_y + c_to_c_h
        return torch.stack((pred_boxes1, pred_boxes2, pred_boxes3, pred_boxes4), dim=2).flatten(1)


class BoxLinearCoder:
    def __init__(self, normalize_by_size: bool = True):
        self.normalize_by_size = normalize_by_size

    def encode(self, reference_boxes: Tensor, proposals: Tensor) -> Tensor:
        ctr_x = 0.5 * (reference_boxes[..., 0] + reference_boxes[..., 2])
        ctr_y = 0.5 * (reference_boxes[..., 1] + reference_boxes[..., 3])
        target_l = ctr_x - proposals[..., 0]
        target_t = ctr_y - proposals[..., 1]
        target_r = proposals[..., 2] - ctr_x
        target_b = proposals[..., 3] - ctr_y
        targets = torch.stack((target_l, target_t, target_r, target_b), dim=-1)
        if self.normalize_by_size:
            w = reference_boxes[..., 2] - reference_boxes[..., 0]
            h = reference_boxes[..., 3] - reference_boxes[..., 1]
            targets = targets / torch.stack((w, h, w, h), dim=-1)
        return targets

    def decode(self, rel_codes: Tensor, boxes: Tensor) -> Tensor:
        boxes = boxes.to(dtype=rel_codes.dtype)
        ctr_x = 0.5 * (boxes[..., 0] + boxes[..., 2])
        ctr_y = 0.5 * (boxes[..., 1] + boxes[..., 3])
        if self.normalize_by_size:
            boxes_w = boxes[..., 2] - boxes[..., 0]
            boxes_h = boxes[..., 3] - boxes[..., 1]
            rel_codes = rel_codes * torch.stack((boxes_w, boxes_h, boxes_w, boxes_h), dim=-1)
        pred_boxes1 = ctr_x - rel_codes[..., 0]
        pred_boxes2 = ctr_y - rel_codes[..., 1]
        pred_boxes3 = ctr_x + rel_codes[..., 2]
        pred_boxes4 = ctr_y + rel_codes[..., 3]
        return torch.stack((pred_boxes1, pred_boxes2, pred_boxes3, pred_boxes4), dim=-1)


def _topk_min(input: Tensor, orig_kval: int, axis: int) -> int:
    return min(orig_kval, input.size(axis))


def topk_stable(scores: Tensor, k: int) -> Tuple[Tensor, Tensor]:
    """``Tensor.topk(k)`` with the adopted tie rule (descending value, ascending index)."""
    idx = _argsort_desc(scores)[:k]
    return scores[idx], idx


# --------------------------------------------------------------------------------------------
# torchvision/models/detection/anchor_utils.py
# --------------------------------------------------------------------------------------------
class AnchorGenerator:
    def __init__(self, sizes=((128, 256, 512),), aspect_ratios=((0.5, 1.0, 2.0),)):
        if not isinstance(sizes[0], (list, tuple)):
            sizes = tuple((s,) for s in sizes)
        if not isinstance(aspect_ratios[0], (list, tuple)):
            aspect_ratios = (aspect_ratios,) * len(sizes)
        self.sizes = sizes
        self.aspect_ratios = aspect_ratios
        self.cell_anchors = [self.generate_anchors(s, a) for s, a in zip(sizes, aspect_ratios)]

    @staticmethod
    def generate_anchors(scales, aspect_ratios, dtype=torch.float32) -> Tensor:
        scales = torch.as_tensor(scales, dtype=dtype)
        aspect_ratios = torch.as_tensor(aspect_ratios, dtype=dtype)
        h_ratios = torch.sqrt(aspect_ratios)
        w_ratios = 1 / h_ratios
        ws = (w_ratios[:, None] * scales[None, :]).view(-1)
        hs = (h_ratios[:, None] * scales[None, :]).view(-1)
        base_anchors = torch.stack([-ws, -hs, ws, hs], dim=1) / 2
        return base_anchors.round()

    def num_anchors_per_location(self) -> List[int]:
        return [len(s) * len(a) for s, a in zip(self.sizes, self.aspect_ratios)]

    def grid_anchors(self, grid_sizes, strides, dtype=torch.float32) -> List[Tensor]:
        cell_anchors = [c.to(dtype) for c in self.cell_anchors]
        assert len(grid_sizes) == len(strides) == len(cell_anchors)
        anchors = []
        for size, stride, base_anchors in zip(grid_sizes, strides, cell_anchors):
            grid_height, grid_width = size
            stride_height, stride_width = stride
            shifts_x = torch.arange(0, grid_width, dtype=torch.int32) * stride_width
            shifts_y = torch.arange(0, grid_height, dtype=torch.int32) * stride_height
            shift_y, shift_x = torch.meshgrid(shifts_y, shifts_x, indexing="ij")
            shift_x = shift_x.reshape(-1)
            shift_y = shift_y.reshape(-1)
            shifts = torch.stack((shift_x, shift_y, shift_x, shift_y), dim=1)
            anchors.append((shifts.view(-1, 1, 4) + base_anchors.view(1, -1, 4)).reshape(-1, 4))
        return anchors

    def forward(self, image_size: Tuple[int, int], grid_sizes: Sequence[Tuple[int, int]],
                num_images: int, dtype=torch.float32) -> List[Tensor]:
        """``image_size`` = padded batch tensor H,W (``image_list.tensors.shape[-2:]``)."""
        strides = [[image_size[0] // g[0], image_size[1] // g[1]] for g in grid_sizes]
        per_level = self.grid_anchors(grid_sizes, strides, dtype)
        return [torch.cat(per_level) for _ in range(num_images)]


# --------------------------------------------------------------------------------------------
# torchvision/ops/focal_loss.py, giou_loss.py (head loss glue; downstream of matching)
# --------------------------------------------------------------------------------------------
def sigmoid_focal_loss(inputs, targets, alpha: float = 0.25, gamma: float = 2, reduction: str = "none"):
    p = torch.sigmoid(inputs)
    ce_loss = torch.nn.functional.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce_loss * ((1 - p_t) ** gamma)
    if alpha >= 0:
        alpha_t = alpha * targets + (1 - alpha) * (1 - targets)
        loss = alpha_t * loss
    if reduction == "mean":
        loss = loss.mean()
    elif reduction == "sum":
        loss = loss.sum()
    return loss


def _loss_inter_union(boxes1, boxes2):
    x1, y1, x2, y2 = boxes1.unbind(dim=-1)
    x1g, y1g, x2g, y2g = boxes2.unbind(dim=-1)
    xkis1 = torch.max(x1, x1g)
    ykis1 = torch.max(y1, y1g)
    xkis2 = torch.min(x2, x2g)
    ykis2 = torch.min(y2, y2g)
    intsctk = torch.zeros_like(x1)
    mask = (ykis2 > ykis1) & (xkis2 > xkis1)
    intsctk[mask] = (xkis2[mask] - xkis1[mask]) * (ykis2[mask] - ykis1[mask])
    unionk = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - intsctk
    return intsctk, unionk


def generalized_box_iou_loss(boxes1, boxes2, reduction: str = "none", eps: float = 1e-7):
    intsctk, unionk = _loss_inter_union(boxes1, boxes2)
    iouk = intsctk / (unionk + eps)
    x1, y1, x2, y2 = boxes1.unbind(dim=-1)
    x1g, y1g, x2g, y2g = boxes2.unbind(dim=-1)
    xc1 = torch.min(x1, x1g)
    yc1 = torch.min(y1, y1g)
    xc2 = torch.max(x2, x2g)
    yc2 = torch.max(y2, y2g)
    area_c = (xc2 - xc1) * (yc2 - yc1)
    miouk = iouk - ((area_c - unionk) / (area_c + eps))
    loss = 1 - miouk
    if reduction == "mean":
        loss = loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
    elif reduction == "sum":
        loss = loss.sum()
    return loss


# --------------------------------------------------------------------------------------------
# torchvision/models/detection/retinanet.py
# --------------------------------------------------------------------------------------------
def retinanet_match(targets_boxes: List[Tensor], anchors: List[Tensor], fg_iou_thresh: float,
                    bg_iou_thresh: float) -> List[Tensor]:
    """Matching half of ``RetinaNet.compute_loss``: per image box_iou -> Matcher(hi, lo, True)."""
    matcher = Matcher(fg_iou_thresh, bg_iou_thresh, allow_low_quality_matches=True)
    out = []
    for anchors_per_image, gt in zip(anchors, targets_boxes):
        if gt.numel() == 0:
            out.append(torch.full((anchors_per_image.size(0),), -1, dtype=torch.int64))
            continue
        out.append(matcher(box_iou(gt, anchors_per_image)))
    return out


def retinanet_match_chunked(gt: Tensor, anchors: Tensor, fg_iou_thresh: float, bg_iou_thresh: float,
                            chunk: int = 65536) -> Tensor:
    """Value-identical to ``Matcher(hi, lo, True)(box_iou(gt, anchors))`` without materialising the
    whole [M,N] matrix: anchors are processed in chunks with a running per-GT maximum, then a second
    sweep applies the low-quality restore.  Used when M*N*4 bytes would not fit (M=2000, N=786k)."""
    n = anchors.size(0)
    if gt.numel() == 0:
        return torch.full((n,), -1, dtype=torch.int64)
    vals = torch.empty(n, dtype=torch.float32)
    all_matches = torch.empty(n, dtype=torch.int64)
    gt_max = torch.full((gt.size(0),), -float("inf"), dtype=torch.float32)
    for s in range(0, n, chunk):
        iou = box_iou(gt, anchors[s:s + chunk])
        v, m = iou.max(dim=0)
        vals[s:s + chunk] = v
        all_matches[s:s + chunk] = m
        gt_max = torch.maximum(gt_max, iou.max(dim=1)[0])
    matches = all_matches.clone()
    matches[vals < bg_iou_thresh] = BELOW_LOW_THRESHOLD
    matches[(vals >= bg_iou_thresh) & (vals < fg_iou_thresh)] = BETWEEN_THRESHOLDS
    for s in range(0, n, chunk):
        iou = box_iou(gt, anchors[s:s + chunk])
        upd = torch.where(iou == gt_max[:, None])[1] + s
        matches[upd] = all_matches[upd]
    return matches


def retinanet_head_loss_regression(targets, bbox_regression, anchors, matched_idxs, loss_type: str = "giou",
                                   beta: float = 1.0, coder: "BoxCoder" = None, eps: float = None):
    """RetinaNetRegressionHead.compute_loss: _box_loss over the foreground anchors / max(1, #fg), batch mean."""
    coder = BoxCoder() if coder is None else coder
    reg_losses = []
    for t, reg, a, m in zip(targets, bbox_regression, anchors, matched_idxs):
        fg = torch.where(m >= 0)[0]
        num_fg = fg.numel()
        gt_boxes = t["boxes"][m[fg]]
        reg_fg = reg[fg, :]
        a_fg = a[fg, :]
        if loss_type == "giou":
            pred = coder.decode_single(reg_fg, a_fg)
            e = torch.finfo(reg_fg.dtype).eps if eps is None else eps
            loss = generalized_box_iou_loss(pred, gt_boxes, reduction="sum", eps=e)
        elif loss_type == "l1":
            loss = torch.nn.functional.l1_loss(reg_fg, coder.encode_single(gt_boxes, a_fg), reduction="sum")
        elif loss_type == "smooth_l1":
            loss = torch.nn.functional.smooth_l1_loss(reg_fg, coder.encode_single(gt_boxes, a_fg), reduction="sum",
                                                      beta=beta)
        else:
            raise NotImplementedError(loss_type)
        reg_losses.append(loss / max(1, num_fg))
    return sum(reg_losses) / max(1, len(targets))


def retinanet_head_loss(targets, head_outputs, anchors, matched_idxs, loss_type: str = "giou", beta: float = 1.0,
                        coder: "BoxCoder" = None):
    """RetinaNetHead.compute_loss = classification (sigmoid focal, ignore -2) + regression (_box_loss)."""
    cls_losses = []
    for t, cls_logits, m in zip(targets, head_outputs["cls_logits"], matched_idxs):
        fg = m >= 0
        num_fg = fg.sum()
        gt_classes_target = torch.zeros_like(cls_logits)
        gt_classes_target[fg, t["labels"][m[fg]]] = 1.0
        valid = m != BETWEEN_THRESHOLDS
        cls_losses.append(sigmoid_focal_loss(cls_logits[valid], gt_classes_target[valid], reduction="sum")
                          / max(1, num_fg))
    return {
        "classification": sum(cls_losses) / len(targets),
        "bbox_regression": retinanet_head_loss_regression(targets, head_outputs["bbox_regression"], anchors, matched_idxs,
                                                          loss_type, beta, coder),
    }


def retinanet_postprocess(head_outputs: Dict[str, List[Tensor]], anchors: List[List[Tensor]],
                          image_shapes: List[Tuple[int, int]], score_thresh: float, topk_candidates: int,
                          nms_thresh: float, detections_per_img: int, math: str = "spec",
                          device_rule: str = "cpu", return_internals: bool = False):
    """``RetinaNet.postprocess_detections``.  head_outputs[k] = list over levels of [B, N_l, .]."""
    class_logits = head_outputs["cls_logits"]
    box_regression = head_outputs["bbox_regression"]
    coder = BoxCoder()
    detections = []
    for index in range(len(image_shapes)):
        box_regression_per_image = [br[index] for br in box_regression]
        logits_per_image = [cl[index] for cl in class_logits]
        anchors_per_image, image_shape = anchors[index], image_shapes[index]
        image_boxes, image_scores, image_labels = [], [], []
        for reg_l, logits_l, anchors_l in zip(box_regression_per_image, logits_per_image, anchors_per_image):
            num_classes = logits_l.shape[-1]
            scores_l = _sigmoid(logits_l, math).flatten()
            keep_idxs = scores_l > score_thresh
            scores_l = scores_l[keep_idxs]
            topk_idxs = torch.where(keep_idxs)[0]
            num_topk = _topk_min(topk_idxs, topk_candidates, 0)
            scores_l, idxs = topk_stable(scores_l, num_topk)
            topk_idxs = topk_idxs[idxs]
            anchor_idxs = torch.div(topk_idxs, num_classes, rounding_mode="floor")
            labels_l = topk_idxs % num_classes
            boxes_l = coder.decode_single(reg_l[anchor_idxs], anchors_l[anchor_idxs], math=math)
            boxes_l = clip_boxes_to_image(boxes_l, image_shape)
            image_boxes.append(boxes_l)
            image_scores.append(scores_l)
            image_labels.append(labels_l)
        image_boxes = torch.cat(image_boxes, dim=0)
        image_scores = torch.cat(image_scores, dim=0)
        image_labels = torch.cat(image_labels, dim=0)
        keep = batched_nms(image_boxes, image_scores, image_labels, nms_thresh, device_rule)
        keep = keep[:detections_per_img]
        det = {"boxes": image_boxes[keep], "scores": image_scores[keep], "labels": image_labels[keep]}
        if return_internals:
            det.update(keep=keep, cand_boxes=image_boxes, cand_scores=image_scores, cand_labels=image_labels)
        detections.append(det)
    return detections


# --------------------------------------------------------------------------------------------
# torchvision/models/detection/transform.py (the reference configures it through
# get_model(min_size=size, max_size=size), object_detection/train.py:142-144)
# --------------------------------------------------------------------------------------------
def resize_boxes(boxes: Tensor, original_size: Sequence[int], new_size: Sequence[int]) -> Tensor:
    ratios = [torch.tensor(s, dtype=torch.float32, device=boxes.device)
              / torch.tensor(s_orig, dtype=torch.float32, device=boxes.device)
              for s, s_orig in zip(new_size, original_size)]
    ratio_height, ratio_width = ratios
    xmin, ymin, xmax, ymax = boxes.unbind(1)
    xmin = xmin * ratio_width
    xmax = xmax * ratio_width
    ymin = ymin * ratio_height
    ymax = ymax * ratio_height
    return torch.stack((xmin, ymin, xmax, ymax), dim=1)


def _resize_image(image: Tensor, self_min_size: float, self_max_size: float, fixed_size=None) -> Tensor:
    """``_resize_image_and_masks`` in eager mode: ``im_shape = image.shape[-2:]`` and the scale factor in Python floats
    ("the normal way"); the fp32-tensor arithmetic belongs to the scripted / traced branch, which the reference's
    train.py / predict.py never take."""
    im_shape = image.shape[-2:]
    size = None
    scale_factor = None
    recompute_scale_factor = None
    if fixed_size is not None:
        size = [fixed_size[1], fixed_size[0]]
    else:
        min_size = min(im_shape)
        max_size = max(im_shape)
        scale_factor = min(self_min_size / min_size, self_max_size / max_size)
        recompute_scale_factor = True
    return torch.nn.functional.interpolate(image[None], size=size, scale_factor=scale_factor, mode="bilinear",
                                           recompute_scale_factor=recompute_scale_factor, align_corners=False)[0]


class GeneralizedRCNNTransform:
    """``GeneralizedRCNNTransform.forward`` in eval or train mode with one ``min_size`` (what the reference passes):
    normalize -> bilinear resize (``recompute_scale_factor=True``) with ``resize_boxes`` of the targets -> zero-padded
    batch whose sides are multiples of ``size_divisible``; ``postprocess`` rescales detections."""

    def __init__(self, min_size: int, max_size: int, image_mean, image_std, size_divisible: int = 32, fixed_size=None):
        self.min_size, self.max_size = min_size, max_size
        self.image_mean, self.image_std = image_mean, image_std
        self.size_divisible, self.fixed_size = size_divisible, fixed_size

    def normalize(self, image: Tensor) -> Tensor:
        dtype, device = image.dtype, image.device
        mean = torch.as_tensor(self.image_mean, dtype=dtype, device=device)
        std = torch.as_tensor(self.image_std, dtype=dtype, device=device)
        return (image - mean[:, None, None]) / std[:, None, None]

    def resize(self, image: Tensor, target=None):
        h, w = image.shape[-2:]
        image = _resize_image(image, float(self.min_size), float(self.max_size), self.fixed_size)
        if target is None:
            return image, target
        target = dict(target)
        target["boxes"] = resize_boxes(target["boxes"], (h, w), image.shape[-2:])
        return image, target

    def batch_images(self, images: List[Tensor]) -> Tensor:
        max_size = [max(s) for s in zip(*[list(img.shape) for img in images])]
        stride = float(self.size_divisible)
        max_size[1] = int(math.ceil(float(max_size[1]) / stride) * stride)
        max_size[2] = int(math.ceil(float(max_size[2]) / stride) * stride)
        batched = images[0].new_full([len(images)] + max_size, 0)
        for i in range(batched.shape[0]):
            img = images[i]
            batched[i, : img.shape[0], : img.shape[1], : img.shape[2]].copy_(img)
        return batched

    def forward(self, images: List[Tensor], targets=None):
        images = [img for img in images]
        targets = None if targets is None else [dict(t) for t in targets]
        for i in range(len(images)):
            image = self.normalize(images[i])
            image, t = self.resize(image, None if targets is None else targets[i])
            images[i] = image
            if targets is not None:
                targets[i] = t
        image_sizes = [(int(img.shape[-2]), int(img.shape[-1])) for img in images]
        return self.batch_images(images), image_sizes, targets

    @staticmethod
    def postprocess(result, image_shapes, original_image_sizes):
        out = []
        for pred, im_s, o_im_s in zip(result, image_shapes, original_image_sizes):
            pred = dict(pred)
            pred["boxes"] = resize_boxes(pred["boxes"], im_s, o_im_s)
            out.append(pred)
        return out


# --------------------------------------------------------------------------------------------
# torchvision/models/detection/fcos.py
# --------------------------------------------------------------------------------------------
def fcos_match(targets_boxes: List[Tensor], anchors: List[Tensor], num_anchors_per_level: List[int],
               center_sampling_radius: float = 1.5) -> List[Tensor]:
    """Matching half of ``FCOS.compute_loss`` (centre sampling, scale ranges, min-area GT)."""
    out = []
    for anchors_per_image, gt_boxes in zip(anchors, targets_boxes):
        if gt_boxes.numel() == 0:
            out.append(torch.full((anchors_per_image.size(0),), -1, dtype=torch.int64))
            continue
        gt_centers = (gt_boxes[:, :2] + gt_boxes[:, 2:]) / 2
        anchor_centers = (anchors_per_image[:, :2] + anchors_per_image[:, 2:]) / 2
        anchor_sizes = anchors_per_image[:, 2] - anchors_per_image[:, 0]
        pairwise_match = (anchor_centers[:, None, :] - gt_centers[None, :, :]).abs_().max(dim=2).values \
            < center_sampling_radius * anchor_sizes[:, None]
        x, y = anchor_centers.unsqueeze(dim=2).unbind(dim=1)
        x0, y0, x1, y1 = gt_boxes.unsqueeze(dim=0).unbind(dim=2)
        pairwise_dist = torch.stack([x - x0, y - y0, x1 - x, y1 - y], dim=2)
        pairwise_match &= pairwise_dist.min(dim=2).values > 0
        lower_bound = anchor_sizes * 4
        lower_bound[: num_anchors_per_level[0]] = 0
        upper_bound = anchor_sizes * 8
        upper_bound[-num_anchors_per_level[-1]:] = float("inf")
        pairwise_dist = pairwise_dist.max(dim=2).values
        pairwise_match &= (pairwise_dist > lower_bound[:, None]) & (pairwise_dist < upper_bound[:, None])
        gt_areas = (gt_boxes[:, 2] - gt_boxes[:, 0]) * (gt_boxes[:, 3] - gt_boxes[:, 1])
        pairwise_match = pairwise_match.to(torch.float32) * (1e8 - gt_areas[None, :])
        min_values, matched_idx = pairwise_match.max(dim=1)
        matched_idx[min_values < 1e-5] = -1
        out.append(matched_idx)
    return out


def fcos_head_loss(targets, head_outputs, anchors, matched_idxs):
    """torchvision FCOSHead.compute_loss: focal classification, GIoU of the linearly decoded boxes and BCE of the
    centre-ness against its encoded target, each summed over the batch and divided by max(1, #foreground)."""
    cls_logits = head_outputs["cls_logits"]
    bbox_regression = head_outputs["bbox_regression"]
    bbox_ctrness = head_outputs["bbox_ctrness"]
    all_gt_classes_targets, all_gt_boxes_targets = [], []
    for targets_per_image, matched_idxs_per_image in zip(targets, matched_idxs):
        if len(targets_per_image["labels"]) == 0:
            gt_classes_targets = targets_per_image["labels"].new_zeros((len(matched_idxs_per_image),))
            gt_boxes_targets = targets_per_image["boxes"].new_zeros((len(matched_idxs_per_image), 4))
        else:
            gt_classes_targets = targets_per_image["labels"][matched_idxs_per_image.clip(min=0)]
            gt_boxes_targets = targets_per_image["boxes"][matched_idxs_per_image.clip(min=0)]
        gt_classes_targets[matched_idxs_per_image < 0] = -1
        all_gt_classes_targets.append(gt_classes_targets)
        all_gt_boxes_targets.append(gt_boxes_targets)
    all_gt_boxes_targets = torch.stack(all_gt_boxes_targets)
    all_gt_classes_targets = torch.stack(all_gt_classes_targets)
    anchors = torch.stack(anchors)
    foregroud_mask = all_gt_classes_targets >= 0
    num_foreground = foregroud_mask.sum().item()
    gt_classes_targets = torch.zeros_like(cls_logits)
    gt_classes_targets[foregroud_mask, all_gt_classes_targets[foregroud_mask]] = 1.0
    loss_cls = sigmoid_focal_loss(cls_logits, gt_classes_targets, reduction="sum")
    coder = BoxLinearCoder(normalize_by_size=True)
    pred_boxes = coder.decode(bbox_regression, anchors)
    loss_bbox_reg = generalized_box_iou_loss(pred_boxes[foregroud_mask], all_gt_boxes_targets[foregroud_mask],
                                             reduction="sum")
    bbox_reg_targets = coder.encode(anchors, all_gt_boxes_targets)
    if len(bbox_reg_targets) == 0:
        gt_ctrness_targets = bbox_reg_targets.new_zeros(bbox_reg_targets.size()[:-1])
    else:
        left_right = bbox_reg_targets[:, :, [0, 2]]
        top_bottom = bbox_reg_targets[:, :, [1, 3]]
        gt_ctrness_targets = torch.sqrt((left_right.min(dim=-1)[0] / left_right.max(dim=-1)[0])
                                        * (top_bottom.min(dim=-1)[0] / top_bottom.max(dim=-1)[0]))
    pred_centerness = bbox_ctrness.squeeze(dim=2)
    loss_bbox_ctrness = torch.nn.functional.binary_cross_entropy_with_logits(
        pred_centerness[foregroud_mask], gt_ctrness_targets[foregroud_mask], reduction="sum")
    return {
        "classification": loss_cls / max(1, num_foreground),
        "bbox_regression": loss_bbox_reg / max(1, num_foreground),
        "bbox_ctrness": loss_bbox_ctrness / max(1, num_foreground),
    }


def fcos_postprocess(head_outputs: Dict[str, List[Tensor]], anchors: List[List[Tensor]],
                     image_shapes: List[Tuple[int, int]], score_thresh: float, topk_candidates: int,
                     nms_thresh: float, detections_per_img: int, math: str = "spec",
                     device_rule: str = "cpu", return_internals: bool = False):
    """``FCOS.postprocess_detections``: score = sqrt(sigmoid(cls) * sigmoid(ctrness))."""
    class_logits = head_outputs["cls_logits"]
    box_regression = head_outputs["bbox_regression"]
    box_ctrness = head_outputs["bbox_ctrness"]
    coder = BoxLinearCoder(normalize_by_size=True)
    detections = []
    for index in range(len(image_shapes)):
        reg_pi = [br[index] for br in box_regression]
        log_pi = [cl[index] for cl in class_logits]
        ctr_pi = [bc[index] for bc in box_ctrness]
        anchors_per_image, image_shape = anchors[index], image_shapes[index]
        image_boxes, image_scores, image_labels = [], [], []
        for reg_l, logits_l, ctr_l, anchors_l in zip(reg_pi, log_pi, ctr_pi, anchors_per_image):
            num_classes = logits_l.shape[-1]
            scores_l = _sqrt(_sigmoid(logits_l, math) * _sigmoid(ctr_l, math), math).flatten()
            keep_idxs = scores_l > score_thresh
            scores_l = scores_l[keep_idxs]
            topk_idxs = torch.where(keep_idxs)[0]
            num_topk = _topk_min(topk_idxs, topk_candidates, 0)
            scores_l, idxs = topk_stable(scores_l, num_topk)
            topk_idxs = topk_idxs[idxs]
            anchor_idxs = torch.div(topk_idxs, num_classes, rounding_mode="floor")
            labels_l = topk_idxs % num_classes
            boxes_l = coder.decode(reg_l[anchor_idxs], anchors_l[anchor_idxs])
            boxes_l = clip_boxes_to_image(boxes_l, image_shape)
            image_boxes.append(boxes_l)
            image_scores.append(scores_l)
            image_labels.append(labels_l)
        image_boxes = torch.cat(image_boxes, dim=0)
        image_scores = torch.cat(image_scores, dim=0)
        image_labels = torch.cat(image_labels, dim=0)
        keep = batched_nms(image_boxes, image_scores, image_labels, nms_thresh, device_rule)
        keep = keep[:detections_per_img]
        det = {"boxes": image_boxes[keep], "scores": image_scores[keep], "labels": image_labels[keep]}
        if return_internals:
            det.update(keep=keep, cand_boxes=image_boxes, cand_scores=image_scores, cand_labels=image_labels)
        detections.append(det)
    return detections


# --------------------------------------------------------------------------------------------
# reference object_detection/predict.py:63-76 -- what the prediction loop does with one image's detections
# --------------------------------------------------------------------------------------------
def predict_tail(prediction: Dict[str, Tensor], image_wh: Tuple[int, int], model_wh: Tuple[int, int],
                 iou_threshold: float = 0.3, float_rule: bool = False):
    """Restatement of reference ``object_detection/predict.py:66-76`` for one image: ``ratio = torch.tensor([w / width,
    h / height] * 2)`` (Python doubles rounded to fp32), ``boxes.mul_(ratio)`` (in place: the prediction dictionary
    keeps the rescaled boxes), ``nms(boxes, scores, 0.3)``, ``boxes.to(torch.int32)`` and the gathers by the kept
    indices.  Returns (boxes int32 [k, 4], scores [k], labels [k], keep [k])."""
    w, h = image_wh
    width, height = model_wh
    boxes, scores, labels = prediction["boxes"], prediction["scores"], prediction["labels"]
    ratio = torch.tensor([w / width, h / height] * 2)
    boxes.mul_(ratio)
    nms_indices = nms(boxes, scores, iou_threshold, float_rule)
    boxes = boxes.to(torch.int32)
    return boxes[nms_indices], scores[nms_indices], labels[nms_indices], nms_indices
