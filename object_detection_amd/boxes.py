"""Box operators with the call signatures of ``torchvision.ops.boxes`` (0.16.2), backed by HIP kernels.

The reference reaches these through torchvision: ``nms`` directly at
``object_detection/predict.py:12,72``; ``box_iou`` / ``clip_boxes_to_image`` / ``batched_nms`` inside
``RetinaNet.compute_loss`` / ``postprocess_detections`` (``object_detection/train.py:62,84``).
"""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor

from . import _C, kernels


def box_area(boxes: Tensor) -> Tensor:
    """(x2 - x1) * (y2 - y1); half precision is upcast to fp32 first, as torchvision does."""
    if boxes.is_floating_point() and boxes.dtype not in (torch.float32, torch.float64):
        boxes = boxes.float()
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def box_iou(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    """Pairwise IoU [len(boxes1), len(boxes2)], fp32, same formula and rounding as torchvision."""
    return kernels.box_iou(boxes1, boxes2)


def clip_boxes_to_image(boxes: Tensor, size: Tuple[int, int]) -> Tensor:
    """Clamp x to [0, width] and y to [0, height] (elementwise; the pipeline fuses this into the
    candidate-selection kernel, this standalone form is a convenience for callers)."""
    height, width = size
    out = boxes.clone()
    out[..., 0::2] = boxes[..., 0::2].clamp(min=0, max=width)
    out[..., 1::2] = boxes[..., 1::2].clamp(min=0, max=height)
    return out


# What torchvision does for tensors on a GPU, which is where these operators run: batched_nms takes the per-class
# branch when numel > 20000 (4000 on the CPU), and the native CUDA kernel compares the fp32 IoU with the threshold
# rounded to float (the CPU kernel promotes the IoU to double).  ``threshold_rule="cpu"`` selects the CPU kernel's
# comparison instead; the two differ only for an IoU that equals float(iou_threshold).
def _threshold_flag(threshold_rule: str) -> int:
    if threshold_rule not in ("cuda", "cpu"):
        raise ValueError(f"threshold_rule must be 'cuda' or 'cpu', got {threshold_rule!r}")
    return _C.NMS_THRESHOLD_FLOAT if threshold_rule == "cuda" else 0


def _matrix_flag(n: int) -> int:
    """These calls keep every survivor, which is what the strip kernel is slow at (it tests every strip against the
    whole kept list); the chip-wide bit-matrix form is faster from a few hundred boxes on (600 boxes: 87 -> 38 us,
    4145: 2.2 ms -> 0.15 ms) and takes up to 8192 boxes."""
    return _C.NMS_PREFER_MATRIX if 256 < n <= _C.NMS_MATRIX_MAX_BOXES else 0


def nms(boxes: Tensor, scores: Tensor, iou_threshold: float, threshold_rule: str = "cuda") -> Tensor:
    """Indices kept by greedy NMS, sorted by decreasing score (stable), int64."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    cnt = kernels._device_ints([n], boxes.device).reshape(1, 1)
    keep, _, _, _, count = kernels.batched_nms_padded(boxes.reshape(1, n, 4), scores.reshape(1, n), None, cnt, n,
                                                      False, float(iou_threshold),
                                                      _C.NMS_TRICK | _threshold_flag(threshold_rule) | _matrix_flag(n), n,
                                                      want_labels=False)
    return keep[0, : int(count.item())]


def batched_nms(boxes: Tensor, scores: Tensor, idxs: Tensor, iou_threshold: float, mode: int = None,
                threshold_rule: str = "cuda") -> Tensor:
    """Per-class NMS.  ``mode`` overrides torchvision's size rule (``_C.NMS_TRICK`` / ``_C.NMS_VANILLA``)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    cnt = kernels._device_ints([n], boxes.device).reshape(1, 1)
    mode = (_C.NMS_AUTO_GPU if mode is None else (mode & 0xff)) | _threshold_flag(threshold_rule) | _matrix_flag(n)
    keep, _, _, _, count = kernels.batched_nms_padded(boxes.reshape(1, n, 4), scores.reshape(1, n),
                                                      idxs.reshape(1, n), cnt, n, False, float(iou_threshold), mode,
                                                      n, want_labels=False)
    return keep[0, : int(count.item())]
