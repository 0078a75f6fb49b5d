"""What the reference's prediction loop does with the detections of a batch
(``object_detection/predict.py:63-76``), as one launch for all images.

The reference handles every image of a batch separately: ``ratio = torch.tensor([w / width, h / height] * 2)``,
``boxes.mul_(ratio)``, ``nms(boxes, scores, 0.3)`` (class-agnostic, on the rescaled boxes), ``boxes.to(torch.int32)``
and three gathers -- six small launches and a device-to-host wait inside ``nms`` per image.  ``rescale_nms`` takes the
list ``model(images)`` returned and does all of it in ``od_predict_tail``; nothing waits for the GPU until somebody
looks at a result.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
from torch import Tensor

from . import _C, kernels
from .retinanet import Detections, _PendingBatch


def _ratios(image_sizes: Sequence[Tuple[int, int]], model_size: Tuple[int, int], device: torch.device) -> Tensor:
    """[B, 4] fp32: (w / width, h / height, w / width, h / height) per image, the quotients taken in Python doubles
    and rounded to fp32 once, as ``torch.tensor([...])`` does (reference predict.py:70)."""
    width, height = model_size
    flat: List[float] = []
    for w, h in image_sizes:
        flat.extend([w / width, h / height] * 2)
    return kernels._device_floats(flat, device).view(-1, 4)


def _padded(predictions: Sequence[Dict[str, Tensor]]):
    """(boxes [B, D, 4], scores [B, D], labels [B, D], counts [B], shared) of a list of detections.  ``shared`` is True
    when the list is exactly the un-materialised output of one ``postprocess_detections`` call: its padded tensors
    are then used where they are, and nobody has to know the counts on the host."""
    first = predictions[0]
    pending = getattr(first, "_pending", None) if isinstance(first, Detections) else None
    if pending is not None and len(predictions) == pending.scores.shape[0] and all(
            isinstance(p, Detections) and p._pending is pending and p._index == i and p._scale is None
            for i, p in enumerate(predictions)):
        return pending.boxes, pending.scores, pending.labels, pending.counts_dev, True
    boxes = [p["boxes"] for p in predictions]
    dev = boxes[0].device
    counts = [int(b.shape[0]) for b in boxes]
    pad = torch.nn.utils.rnn.pad_sequence
    return (pad(boxes, batch_first=True).float().contiguous(),
            pad([p["scores"] for p in predictions], batch_first=True).float().contiguous(),
            pad([p["labels"] for p in predictions], batch_first=True).to(torch.int64).contiguous(),
            kernels._device_ints(counts, dev), False)


def rescale_nms(predictions: Sequence[Dict[str, Tensor]], image_sizes: Sequence[Tuple[int, int]],
                model_size: Tuple[int, int], iou_threshold: float = 0.3,
                nms_mode: int = _C.NMS_THRESHOLD_FLOAT) -> List[Dict[str, Tensor]]:
    """The per-image tail of the reference's prediction loop for a whole batch.

    ``predictions``: the list ``model(images)`` returned; ``image_sizes``: ``(w, h)`` of every original image (PIL's
    ``img.size``, reference predict.py:66); ``model_size``: ``(width, height)`` stored with the checkpoint (reference
    predict.py:51-52).  Like the reference, the boxes of ``predictions`` are rescaled in place.  Returns one dictionary
    per image with ``boxes`` int32 [k, 4], ``scores`` [k], ``labels`` [k] -- what the reference draws.
    ``nms_mode``: the threshold rule of torchvision's CUDA kernel by default (the reference predicts on ``cuda``,
    predict.py:29); 0 selects the rule of its CPU kernel."""
    if len(predictions) == 0:
        return []
    if len(image_sizes) != len(predictions):
        raise ValueError(f"rescale_nms: {len(predictions)} predictions but {len(image_sizes)} image sizes")
    boxes, scores, labels, counts, shared = _padded(predictions)
    kernels._require_gpu(boxes, "boxes")
    ratio = _ratios(image_sizes, model_size, boxes.device)
    if boxes.shape[1] == 0:
        out = (torch.zeros((len(predictions), 0, 4), dtype=torch.int32, device=boxes.device), scores, labels,
               torch.zeros((len(predictions),), dtype=torch.int32, device=boxes.device))
    else:
        out = kernels.predict_tail(boxes.contiguous(), scores.contiguous(), labels.contiguous(), counts, ratio,
                                   iou_threshold, nms_mode, scale_in_place=shared and boxes.is_contiguous(),
                                   want_keep_idx=False)[1:]
        if not (shared and boxes.is_contiguous()):
            # the same fp32 products, written into the callers' own tensors (one multi-tensor launch)
            torch._foreach_mul_([p["boxes"] for p in predictions], list(ratio.unbind(0)))
    pending = _PendingBatch(*out)
    return [Detections(pending, i) for i in range(len(predictions))]
