"""Matcher / box coders / loss helpers with the interfaces of
``torchvision.models.detection._utils`` (0.16.2), which the reference's models are built from
(``object_detection/retinanet/retinanet_model.py:4-7``).  Matching, decoding and the head losses (forward and
backward) run in HIP kernels; there is no second implementation in PyTorch ops: inputs the kernels do not take are
rejected with an error that says what they take.
"""
from __future__ import annotations

import math
from typing import List, Tuple

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
    """Centre/size delta coding.  ``decode_single`` runs the HIP kernel (specified fp32 exp) and builds no autograd
    graph: inputs that require a gradient are rejected (the differentiable decode of torchvision's ``_box_loss`` lives
    in the fused loss kernels)."""

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
        """Boxes decoded from fp32 deltas [n, 4] on the GPU (specified fp32 exp; no autograd graph is built: the
        differentiable use of this function inside torchvision's ``_box_loss`` is part of the fused loss kernels)."""
        kernels._require_gpu(rel_codes, "rel_codes")
        if not (rel_codes.dim() == 2 and rel_codes.shape[1] == 4 and rel_codes.dtype is torch.float32):
            raise RuntimeError(f"decode_single takes fp32 deltas of shape [n, 4], got {rel_codes.dtype} {tuple(rel_codes.shape)}")
        if torch.is_grad_enabled() and (rel_codes.requires_grad or boxes.requires_grad):
            raise RuntimeError("decode_single builds no autograd graph; gradients of the box loss come from "
                               "RetinaNetRegressionHead.compute_loss (fused forward and backward kernels)")
        codes = rel_codes
        if tuple(self.weights) != (1.0, 1.0, 1.0, 1.0):
            codes = rel_codes / rel_codes.new_tensor(self.weights)   # same division torchvision performs
        return kernels.decode_boxes(codes, boxes.to(rel_codes.dtype), _C.CODER_BOXCODER, self.bbox_xform_clip)


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
        kernels._require_gpu(rel_codes, "rel_codes")
        if not (self.normalize_by_size and rel_codes.dtype is torch.float32 and rel_codes.shape == boxes.shape):
            raise RuntimeError("BoxLinearCoder.decode takes fp32 codes and boxes of equal shape with normalize_by_size=True")
        if torch.is_grad_enabled() and (rel_codes.requires_grad or boxes.requires_grad):
            raise RuntimeError("BoxLinearCoder.decode builds no autograd graph; gradients of the FCOS box loss come "
                               "from FCOSHead.compute_loss (fused forward and backward kernels)")
        return kernels.decode_boxes(rel_codes, boxes.to(dtype=rel_codes.dtype), _C.CODER_LINEAR, 0.0).reshape(rel_codes.shape)


def _topk_min(input: Tensor, orig_kval: int, axis: int) -> int:
    return min(orig_kval, input.size(axis))


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
                                  normalization: int = _C.LOSS_NORM_PER_IMAGE) -> Tensor:
    """RetinaNetClassificationHead.compute_loss for a batch: logits [B, N, C] on the GPU, fp32 / fp16 / bf16.  16-bit
    logits (a head running under autocast) are read as they are and the loss is evaluated in fp32."""
    kernels._require_gpu(cls_logits, "cls_logits")
    if not (cls_logits.dtype in _FUSED_DTYPES and cls_logits.dim() == 3):
        raise RuntimeError(f"classification loss takes fp32 / fp16 / bf16 logits [B, N, C], got {cls_logits.dtype} "
                           f"{tuple(cls_logits.shape)}")
    if cls_logits.stride(2) != 1 or cls_logits.stride(1) != cls_logits.shape[2]:
        cls_logits = cls_logits.contiguous()
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
                              matched_idxs: List[Tensor], gt_boxes: List[Tensor], cnf=None) -> Tensor:
    """RetinaNetRegressionHead.compute_loss for a batch: regression [B, N, 4] on the GPU, fp32 / fp16 / bf16; anchors
    and targets take no gradient.  Under autocast the regression and the anchors arrive in a 16-bit type: the values
    are used as they are and the loss is evaluated in fp32."""
    torch._assert(type in _BOX_LOSS_TYPES, f"Unsupported loss: {type}")
    kernels._require_gpu(bbox_regression, "bbox_regression")
    if not (bbox_regression.dtype in _FUSED_DTYPES and bbox_regression.dim() == 3 and bbox_regression.shape[2] == 4):
        raise RuntimeError(f"regression loss takes fp32 / fp16 / bf16 deltas [B, N, 4], got {bbox_regression.dtype} "
                           f"{tuple(bbox_regression.shape)}")
    B, n = bbox_regression.shape[0], bbox_regression.shape[1]
    if (bbox_regression.stride(2) != 1 or bbox_regression.stride(1) != 4
            or bbox_regression.stride(0) % 4 or bbox_regression.data_ptr() % (4 * bbox_regression.element_size())):
        bbox_regression = bbox_regression.contiguous()
    if any(a.dtype not in _FUSED_DTYPES or a.shape != (n, 4) for a in anchors):
        raise RuntimeError(f"regression loss takes one [N, 4] floating-point anchor table per image (N = {n})")
    if any(a.requires_grad for a in anchors) or any(g.requires_grad for g in gt_boxes):
        raise RuntimeError("regression loss: anchors and target boxes take no gradient")
    first = anchors[0]
    if all(a is first or a.data_ptr() == first.data_ptr() for a in anchors):
        anchor_table = first.float().contiguous()            # AnchorGenerator hands every image the same tensor
    else:
        anchor_table = torch.stack([a.float() for a in anchors])
    matched = _matched_rows(matched_idxs, n)
    boxes, gt_offsets, _ = kernels._pack_gt(gt_boxes, bbox_regression.device)     # in place when the transform packed them
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
                     matched_idxs: List[Tensor], targets) -> dict:
    """FCOSHead.compute_loss for a batch: {"classification", "bbox_regression", "bbox_ctrness"}.  Head outputs on the
    GPU in one element type (fp32 / fp16 / bf16); anchors and targets take no gradient."""
    kernels._require_gpu(cls_logits, "cls_logits")
    ok = (cls_logits.dim() == 3 and bbox_regression.dim() == 3 and bbox_ctrness.dim() == 3
          and cls_logits.dtype in _FUSED_DTYPES and bbox_regression.dtype is cls_logits.dtype
          and bbox_ctrness.dtype is cls_logits.dtype and bbox_regression.shape[2] == 4 and bbox_ctrness.shape[2] == 1)
    if not ok:
        raise RuntimeError("FCOS head losses take cls_logits [B, N, C], bbox_regression [B, N, 4] and bbox_ctrness "
                           "[B, N, 1] of one element type (fp32 / fp16 / bf16)")
    if any(a.requires_grad for a in anchors) or any(t["boxes"].requires_grad for t in targets):
        raise RuntimeError("FCOS head losses: anchors and target boxes take no gradient")
    if not (bbox_regression.stride(2) == 1 and bbox_regression.stride(1) == 4 and bbox_regression.stride(0) % 4 == 0
            and bbox_regression.data_ptr() % (4 * bbox_regression.element_size()) == 0):
        bbox_regression = bbox_regression.contiguous()
    if bbox_ctrness.stride(1) != 1:
        bbox_ctrness = bbox_ctrness.contiguous()
    n = cls_logits.shape[1]
    if any(a.shape != (n, 4) or a.dtype not in _FUSED_DTYPES for a in anchors):
        raise RuntimeError(f"FCOS head losses take one [N, 4] floating-point anchor table per image (N = {n})")
    cls = retinanet_classification_loss(cls_logits, matched_idxs, [t["labels"] for t in targets],
                                        normalization=_C.LOSS_NORM_BATCH)
    first = anchors[0]
    if all(a is first or a.data_ptr() == first.data_ptr() for a in anchors):
        anchor_table = first.float().contiguous()
    else:
        anchor_table = torch.stack([a.float() for a in anchors])
    matched = _matched_rows(matched_idxs, n)
    boxes, gt_offsets, _ = kernels._pack_gt([t["boxes"] for t in targets], cls_logits.device)
    reg, ctr = _FCOSBoxLoss.apply(bbox_regression, bbox_ctrness, anchor_table, matched, boxes, gt_offsets, 1e-7)
    return {"classification": cls, "bbox_regression": reg, "bbox_ctrness": ctr}
