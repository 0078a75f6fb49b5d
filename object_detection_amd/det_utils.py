"""Matcher / box coders / loss helpers with the interfaces of
``torchvision.models.detection._utils`` (0.16.2), which the reference's models are built from
(``object_detection/retinanet/retinanet_model.py:4-7``).  Index-producing work runs in HIP kernels;
the classification-head loss has a fused
forward/backward kernel pair; the remaining differentiable arithmetic (box losses) stays in PyTorch autograd.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _C, kernels


class Matcher:
    """Assigns to each prediction (column) a ground-truth row, -1 (below ``low_threshold``) or -2
    (between the thresholds); optionally restores low-quality matches."""

    BELOW_LOW_THRESHOLD = -1
    BETWEEN_THRESHOLDS = -2

    def __init__(self, high_threshold: float, low_threshold: float, allow_low_quality_matches: bool = False):
        torch._assert(low_threshold <= high_threshold, "low_threshold should be <= high_threshold")
        self.high_threshold = high_threshold
        self.low_threshold = low_threshold
        self.allow_low_quality_matches = allow_low_quality_matches

    def __call__(self, match_quality_matrix: Tensor) -> Tensor:
        if match_quality_matrix.numel() == 0:
            if match_quality_matrix.shape[0] == 0:
                raise ValueError("No ground-truth boxes available for one of the images during training")
            raise ValueError("No proposal boxes available for one of the images during training")
        return kernels.matcher(match_quality_matrix, self.high_threshold, self.low_threshold,
                               self.allow_low_quality_matches)


class BoxCoder:
    """Centre/size delta coding.  ``decode_single`` without autograd runs the HIP kernel (specified
    fp32 exp); when gradients are required it runs the same formulas as PyTorch ops."""

    def __init__(self, weights: Tuple[float, float, float, float], bbox_xform_clip: float = math.log(1000.0 / 16)):
        self.weights = weights
        self.bbox_xform_clip = bbox_xform_clip

    def encode_single(self, reference_boxes: Tensor, proposals: Tensor) -> Tensor:
        wx, wy, ww, wh = self.weights
        ex_w = proposals[:, 2] - proposals[:, 0]
        ex_h = proposals[:, 3] - proposals[:, 1]
        ex_cx = proposals[:, 0] + 0.5 * ex_w
        ex_cy = proposals[:, 1] + 0.5 * ex_h
        gt_w = reference_boxes[:, 2] - reference_boxes[:, 0]
        gt_h = reference_boxes[:, 3] - reference_boxes[:, 1]
        gt_cx = reference_boxes[:, 0] + 0.5 * gt_w
        gt_cy = reference_boxes[:, 1] + 0.5 * gt_h
        return torch.stack((wx * (gt_cx - ex_cx) / ex_w, wy * (gt_cy - ex_cy) / ex_h,
                            ww * torch.log(gt_w / ex_w), wh * torch.log(gt_h / ex_h)), dim=1)

    def decode_single(self, rel_codes: Tensor, boxes: Tensor) -> Tensor:
        boxes = boxes.to(rel_codes.dtype)
        needs_grad = torch.is_grad_enabled() and (rel_codes.requires_grad or boxes.requires_grad)
        single = rel_codes.dim() == 2 and rel_codes.shape[1] == 4
        if rel_codes.is_cuda and single and not needs_grad and rel_codes.dtype == torch.float32:
            codes = rel_codes
            if tuple(self.weights) != (1.0, 1.0, 1.0, 1.0):
                codes = rel_codes / rel_codes.new_tensor(self.weights)   # same division torchvision performs
            return kernels.decode_boxes(codes, boxes, _C.CODER_BOXCODER, self.bbox_xform_clip)
        widths = boxes[:, 2] - boxes[:, 0]
        heights = boxes[:, 3] - boxes[:, 1]
        ctr_x = boxes[:, 0] + 0.5 * widths
        ctr_y = boxes[:, 1] + 0.5 * heights
        wx, wy, ww, wh = self.weights
        dx = rel_codes[:, 0::4] / wx
        dy = rel_codes[:, 1::4] / wy
        dw = torch.clamp(rel_codes[:, 2::4] / ww, max=self.bbox_xform_clip)
        dh = torch.clamp(rel_codes[:, 3::4] / wh, max=self.bbox_xform_clip)
        pred_ctr_x = dx * widths[:, None] + ctr_x[:, None]
        pred_ctr_y = dy * heights[:, None] + ctr_y[:, None]
        half_w = 0.5 * (torch.exp(dw) * widths[:, None])
        half_h = 0.5 * (torch.exp(dh) * heights[:, None])
        return torch.stack((pred_ctr_x - half_w, pred_ctr_y - half_h, pred_ctr_x + half_w, pred_ctr_y + half_h),
                           dim=2).flatten(1)


class BoxLinearCoder:
    """FCOS distance coding (left, top, right, bottom), optionally normalised by the anchor size."""

    def __init__(self, normalize_by_size: bool = True):
        self.normalize_by_size = normalize_by_size

    def encode(self, reference_boxes: Tensor, proposals: Tensor) -> Tensor:
        ctr_x = 0.5 * (reference_boxes[..., 0] + reference_boxes[..., 2])
        ctr_y = 0.5 * (reference_boxes[..., 1] + reference_boxes[..., 3])
        targets = torch.stack((ctr_x - proposals[..., 0], ctr_y - proposals[..., 1],
                               proposals[..., 2] - ctr_x, proposals[..., 3] - ctr_y), dim=-1)
        if self.normalize_by_size:
            w = reference_boxes[..., 2] - reference_boxes[..., 0]
            h = reference_boxes[..., 3] - reference_boxes[..., 1]
            targets = targets / torch.stack((w, h, w, h), dim=-1)
        return targets

    def decode(self, rel_codes: Tensor, boxes: Tensor) -> Tensor:
        boxes = boxes.to(dtype=rel_codes.dtype)
        needs_grad = torch.is_grad_enabled() and (rel_codes.requires_grad or boxes.requires_grad)
        if (rel_codes.is_cuda and not needs_grad and self.normalize_by_size and rel_codes.dtype == torch.float32
                and rel_codes.shape == boxes.shape):
            return kernels.decode_boxes(rel_codes, boxes, _C.CODER_LINEAR, 0.0).reshape(rel_codes.shape)
        ctr_x = 0.5 * (boxes[..., 0] + boxes[..., 2])
        ctr_y = 0.5 * (boxes[..., 1] + boxes[..., 3])
        if self.normalize_by_size:
            w = boxes[..., 2] - boxes[..., 0]
            h = boxes[..., 3] - boxes[..., 1]
            rel_codes = rel_codes * torch.stack((w, h, w, h), dim=-1)
        return torch.stack((ctr_x - rel_codes[..., 0], ctr_y - rel_codes[..., 1],
                            ctr_x + rel_codes[..., 2], ctr_y + rel_codes[..., 3]), dim=-1)


def _topk_min(input: Tensor, orig_kval: int, axis: int) -> int:
    return min(orig_kval, input.size(axis))


# ---- loss helpers (autograd, PyTorch ops) ---------------------------------------------------------
def sigmoid_focal_loss(inputs: Tensor, targets: Tensor, alpha: float = 0.25, gamma: float = 2,
                       reduction: str = "none") -> Tensor:
    p = torch.sigmoid(inputs)
    ce = torch.nn.functional.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    if reduction == "mean":
        return loss.mean()
    if reduction == "sum":
        return loss.sum()
    return loss


def generalized_box_iou_loss(boxes1: Tensor, boxes2: Tensor, reduction: str = "none", eps: float = 1e-7) -> Tensor:
    x1, y1, x2, y2 = boxes1.unbind(dim=-1)
    x1g, y1g, x2g, y2g = boxes2.unbind(dim=-1)
    xk1, yk1 = torch.max(x1, x1g), torch.max(y1, y1g)
    xk2, yk2 = torch.min(x2, x2g), torch.min(y2, y2g)
    inter = torch.zeros_like(x1)
    mask = (yk2 > yk1) & (xk2 > xk1)
    inter[mask] = (xk2[mask] - xk1[mask]) * (yk2[mask] - yk1[mask])
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
    iou = inter / (union + eps)
    area_c = (torch.max(x2, x2g) - torch.min(x1, x1g)) * (torch.max(y2, y2g) - torch.min(y1, y1g))
    loss = 1 - (iou - ((area_c - union) / (area_c + eps)))
    if reduction == "mean":
        return loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
    if reduction == "sum":
        return loss.sum()
    return loss


def _box_loss(type: str, box_coder: BoxCoder, anchors_per_image: Tensor, matched_gt_boxes_per_image: Tensor,
              bbox_regression_per_image: Tensor, cnf=None) -> Tensor:
    torch._assert(type in ["l1", "smooth_l1", "giou"], f"Unsupported loss: {type}")
    if type == "l1":
        target = box_coder.encode_single(matched_gt_boxes_per_image, anchors_per_image)
        return torch.nn.functional.l1_loss(bbox_regression_per_image, target, reduction="sum")
    if type == "smooth_l1":
        target = box_coder.encode_single(matched_gt_boxes_per_image, anchors_per_image)
        beta = cnf["beta"] if cnf is not None and "beta" in cnf else 1.0
        return torch.nn.functional.smooth_l1_loss(bbox_regression_per_image, target, reduction="sum", beta=beta)
    pred = box_coder.decode_single(bbox_regression_per_image, anchors_per_image)
    eps = torch.finfo(bbox_regression_per_image.dtype).eps
    return generalized_box_iou_loss(pred, matched_gt_boxes_per_image, reduction="sum", eps=eps)


_FUSED_DTYPES = (torch.float32, torch.float16, torch.bfloat16)


class _RetinaNetFocalLoss(torch.autograd.Function):
    """Classification-head loss of a whole batch in one streamed pass each way (kernels.focal_loss_*)."""

    @staticmethod
    def forward(ctx, cls_logits: Tensor, matched: Tensor, gt_labels: Tensor, gt_offsets: Tensor, alpha: float,
                gamma: float, normalization: int = _C.LOSS_NORM_PER_IMAGE) -> Tensor:
        loss, _, num_fg = kernels.focal_loss_forward(cls_logits, matched, gt_labels, gt_offsets, alpha, gamma,
                                                     normalization)
        ctx.save_for_backward(cls_logits, matched, gt_labels, gt_offsets, num_fg)
        ctx.alpha, ctx.gamma, ctx.normalization = alpha, gamma, normalization
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        cls_logits, matched, gt_labels, gt_offsets, num_fg = ctx.saved_tensors
        g = grad_out.reshape(1).to(torch.float32).contiguous()
        grad = kernels.focal_loss_backward(cls_logits, matched, gt_labels, gt_offsets, ctx.alpha, ctx.gamma, num_fg, g,
                                           ctx.normalization)
        return grad, None, None, None, None, None, None


def _matched_rows(matched_idxs: List[Tensor], n: int) -> Tensor:
    """[B, N] int64 view of the per-image matched indices (no copy when they are the rows of one tensor, which is what
    match_anchors returns)."""
    first = matched_idxs[0]
    step = n * first.element_size()
    if first.dtype is torch.int64 and first.is_contiguous() and all(
            m.data_ptr() == first.data_ptr() + i * step and m.shape == first.shape for i, m in enumerate(matched_idxs)):
        return first.as_strided((len(matched_idxs), n), (n, 1))
    return torch.stack([m.to(torch.int64) for m in matched_idxs])


def retinanet_classification_loss(cls_logits: Tensor, matched_idxs: List[Tensor], labels: List[Tensor],
                                  alpha: float = 0.25, gamma: float = 2.0,
                                  normalization: int = _C.LOSS_NORM_PER_IMAGE) -> Optional[Tensor]:
    """Fused path of RetinaNetClassificationHead.compute_loss; None if the inputs do not qualify (not on the GPU,
    not fp32 / fp16 / bf16, or logits whose inner dimensions are not contiguous) and the caller should use the torch
    ops.  16-bit logits (a head running under autocast) are read as they are and the loss is evaluated in fp32."""
    if not (cls_logits.is_cuda and cls_logits.dtype in _FUSED_DTYPES and cls_logits.dim() == 3):
        return None
    if cls_logits.stride(2) != 1 or cls_logits.stride(1) != cls_logits.shape[2]:
        return None
    B, n = cls_logits.shape[0], cls_logits.shape[1]
    matched = _matched_rows(matched_idxs, n)
    offsets = [0]
    for lab in labels:
        offsets.append(offsets[-1] + lab.shape[0])
    if offsets[-1] > 0:
        gt_labels = torch.cat([lab.to(torch.int64) for lab in labels])
    else:
        gt_labels = torch.zeros((1,), dtype=torch.int64, device=cls_logits.device)
    gt_offsets = kernels._device_ints(offsets, cls_logits.device)
    return _RetinaNetFocalLoss.apply(cls_logits, matched, gt_labels, gt_offsets, float(alpha), float(gamma),
                                     int(normalization))


class _RetinaNetBoxLoss(torch.autograd.Function):
    """Regression-head loss of a whole batch: one pass over the matched indices each way (kernels.box_loss_*)."""

    @staticmethod
    def forward(ctx, bbox_regression: Tensor, anchors: Tensor, matched: Tensor, gt_boxes: Tensor, gt_offsets: Tensor,
                loss_type: int, beta: float, weights, clip: float, eps: float) -> Tensor:
        conf = (loss_type, beta, tuple(weights), clip, eps)
        loss, _, num_fg = kernels.box_loss_forward(bbox_regression, anchors, matched, gt_boxes, gt_offsets, *conf)
        ctx.save_for_backward(bbox_regression, anchors, matched, gt_boxes, gt_offsets, num_fg)
        ctx.conf = conf
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        bbox_regression, anchors, matched, gt_boxes, gt_offsets, num_fg = ctx.saved_tensors
        g = grad_out.reshape(1).to(torch.float32).contiguous()
        grad = kernels.box_loss_backward(bbox_regression, anchors, matched, gt_boxes, gt_offsets, *ctx.conf, num_fg, g)
        return (grad,) + (None,) * 9


_BOX_LOSS_TYPES = {"l1": _C.BOX_LOSS_L1, "smooth_l1": _C.BOX_LOSS_SMOOTH_L1, "giou": _C.BOX_LOSS_GIOU}


def retinanet_regression_loss(type: str, box_coder: BoxCoder, bbox_regression: Tensor, anchors: List[Tensor],
                              matched_idxs: List[Tensor], gt_boxes: List[Tensor], cnf=None) -> Optional[Tensor]:
    """Fused path of RetinaNetRegressionHead.compute_loss; None if the inputs do not qualify (not on the GPU, not
    fp32 / fp16 / bf16, rows not contiguous, anchors or targets that need a gradient) and the caller should use the torch
    ops.  Under autocast the regression and the anchors arrive in a 16-bit type: the values are used as they are and
    the loss is evaluated in fp32."""
    torch._assert(type in _BOX_LOSS_TYPES, f"Unsupported loss: {type}")
    if not (bbox_regression.is_cuda and bbox_regression.dtype in _FUSED_DTYPES and bbox_regression.dim() == 3):
        return None
    B, n = bbox_regression.shape[0], bbox_regression.shape[1]
    if (bbox_regression.shape[2] != 4 or bbox_regression.stride(2) != 1 or bbox_regression.stride(1) != 4
            or bbox_regression.stride(0) % 4 or bbox_regression.data_ptr() % (4 * bbox_regression.element_size())):
        return None
    if any(a.requires_grad or a.dtype not in _FUSED_DTYPES or a.shape != (n, 4) for a in anchors):
        return None
    if any(g.requires_grad for g in gt_boxes):
        return None
    first = anchors[0]
    if all(a is first or a.data_ptr() == first.data_ptr() for a in anchors):
        anchor_table = first.float().contiguous()            # AnchorGenerator hands every image the same tensor
    else:
        anchor_table = torch.stack([a.float() for a in anchors])
    matched = _matched_rows(matched_idxs, n)
    offsets = [0]
    for g in gt_boxes:
        offsets.append(offsets[-1] + g.shape[0])
    if offsets[-1] > 0:
        boxes = torch.cat([g.to(torch.float32).reshape(-1, 4) for g in gt_boxes])
    else:
        boxes = torch.zeros((1, 4), dtype=torch.float32, device=bbox_regression.device)
    gt_offsets = kernels._device_ints(offsets, bbox_regression.device)
    beta = float(cnf["beta"]) if cnf is not None and "beta" in cnf else 1.0
    eps = float(torch.finfo(bbox_regression.dtype).eps)  # as _box_loss: the epsilon of the regression's element type
    return _RetinaNetBoxLoss.apply(bbox_regression, anchor_table, matched, boxes, gt_offsets, _BOX_LOSS_TYPES[type], beta,
                                   tuple(float(w) for w in box_coder.weights), float(box_coder.bbox_xform_clip), eps)


class _FCOSBoxLoss(torch.autograd.Function):
    """Regression (GIoU) and centre-ness losses of the FCOS head for a whole batch (kernels.fcos_box_loss_*)."""

    @staticmethod
    def forward(ctx, bbox_regression: Tensor, bbox_ctrness: Tensor, anchors: Tensor, matched: Tensor, gt_boxes: Tensor,
                gt_offsets: Tensor, eps: float):
        loss_reg, loss_ctr, num_fg = kernels.fcos_box_loss_forward(bbox_regression, bbox_ctrness, anchors, matched,
                                                                   gt_boxes, gt_offsets, eps)
        ctx.save_for_backward(bbox_regression, bbox_ctrness, anchors, matched, gt_boxes, gt_offsets, num_fg)
        ctx.eps = eps
        return loss_reg.reshape(()), loss_ctr.reshape(())

    @staticmethod
    def backward(ctx, grad_reg_loss: Tensor, grad_ctr_loss: Tensor):
        bbox_regression, bbox_ctrness, anchors, matched, gt_boxes, gt_offsets, num_fg = ctx.saved_tensors
        g_reg = grad_reg_loss.reshape(1).to(torch.float32).contiguous()
        g_ctr = grad_ctr_loss.reshape(1).to(torch.float32).contiguous()
        grad_reg, grad_ctr = kernels.fcos_box_loss_backward(bbox_regression, bbox_ctrness, anchors, matched, gt_boxes,
                                                            gt_offsets, ctx.eps, num_fg, g_reg, g_ctr)
        return grad_reg, grad_ctr.reshape(bbox_ctrness.shape), None, None, None, None, None


def fcos_head_losses(cls_logits: Tensor, bbox_regression: Tensor, bbox_ctrness: Tensor, anchors: List[Tensor],
                     matched_idxs: List[Tensor], targets) -> Optional[dict]:
    """Fused path of FCOSHead.compute_loss: {"classification", "bbox_regression", "bbox_ctrness"}; None if the inputs
    do not qualify (not on the GPU, element types other than fp32 / fp16 / bf16, rows not contiguous) and the caller
    should use the torch ops."""
    ok = (cls_logits.is_cuda and cls_logits.dim() == 3 and bbox_regression.dim() == 3 and bbox_ctrness.dim() == 3
          and cls_logits.dtype in _FUSED_DTYPES and bbox_regression.dtype is cls_logits.dtype
          and bbox_ctrness.dtype is cls_logits.dtype and bbox_regression.shape[2] == 4 and bbox_ctrness.shape[2] == 1
          and bbox_regression.stride(2) == 1 and bbox_regression.stride(1) == 4 and bbox_regression.stride(0) % 4 == 0
          and bbox_regression.data_ptr() % (4 * bbox_regression.element_size()) == 0 and bbox_ctrness.stride(1) == 1
          and not any(a.requires_grad for a in anchors) and not any(t["boxes"].requires_grad for t in targets))
    if not ok:
        return None
    n = cls_logits.shape[1]
    if any(a.shape != (n, 4) or a.dtype not in _FUSED_DTYPES for a in anchors):
        return None
    cls = retinanet_classification_loss(cls_logits, matched_idxs, [t["labels"] for t in targets],
                                        normalization=_C.LOSS_NORM_BATCH)
    if cls is None:
        return None
    first = anchors[0]
    if all(a is first or a.data_ptr() == first.data_ptr() for a in anchors):
        anchor_table = first.float().contiguous()
    else:
        anchor_table = torch.stack([a.float() for a in anchors])
    matched = _matched_rows(matched_idxs, n)
    offsets = [0]
    for t in targets:
        offsets.append(offsets[-1] + t["boxes"].shape[0])
    if offsets[-1] > 0:
        boxes = torch.cat([t["boxes"].to(torch.float32).reshape(-1, 4) for t in targets])
    else:
        boxes = torch.zeros((1, 4), dtype=torch.float32, device=cls_logits.device)
    gt_offsets = kernels._device_ints(offsets, cls_logits.device)
    reg, ctr = _FCOSBoxLoss.apply(bbox_regression, bbox_ctrness, anchor_table, matched, boxes, gt_offsets, 1e-7)
    return {"classification": cls, "bbox_regression": reg, "bbox_ctrness": ctr}
