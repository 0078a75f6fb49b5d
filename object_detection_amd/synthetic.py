"""Seeded synthetic inputs for the box pipeline (shared by tests, fixtures and bench.py).

Everything is generated on the CPU with ``torch.Generator().manual_seed(1000 * cfg + image_idx)``
and copied to the device afterwards, so CPU checker and HIP kernels see identical bits.
Distributions: ``cls_logits ~ N(-3, 1.5)`` with 0.1 % set to +20 (sigmoid saturates to exactly
1.0, forcing score ties); ``bbox_regression ~ N(0, 0.5)`` with 0.01 % of dw/dh set to 5.0 (above
the ``log(1000/16)`` clamp); GT boxes: centres uniform over the image, sides log-uniform in
[8, 256] px, clipped to the image with a 2 px minimum side.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import torch


@dataclass(frozen=True)
class PipelineConfig:
    name: str
    cfg_id: int
    batch: int
    image_size: Tuple[int, int]
    grids: Tuple[int, ...]                 # square level grids, finest first
    anchors_per_loc: int
    num_classes: int
    topk_candidates: int
    detections_per_img: int
    score_thresh: float
    nms_thresh: float
    num_gt: int = 0
    fg_iou_thresh: float = 0.6
    bg_iou_thresh: float = 0.5

    @property
    def num_anchors(self) -> int:
        return self.anchors_per_loc * sum(g * g for g in self.grids)


# BASELINE.json configs made concrete (square images; grids = image // stride)
CFG1 = PipelineConfig("retinanet_r50fpn_512_b2", 1, 2, (512, 512), (64, 32, 16, 8, 4), 9, 4, 1000, 300, 0.5, 0.3)
CFG2 = PipelineConfig("retinanet_effb4_bifpn_1024_b16", 2, 16, (1024, 1024), (256, 128, 64, 32, 16, 8), 9, 4,
                      1000, 500, 0.5, 0.3, num_gt=100)
CFG3 = PipelineConfig("fcos_r50fpn_1024_b16_m200", 3, 16, (1024, 1024), (128, 64, 32, 16, 8), 1, 4, 1000, 100,
                      0.2, 0.6, num_gt=200)
CFG4 = PipelineConfig("effdet_d7_1536_c91", 4, 8, (1536, 1536), (384, 192, 96, 48, 24, 12), 9, 91, 1000, 100,
                      0.05, 0.5)
CFG5 = PipelineConfig("retinanet_dense_match_1024_b64_m2000", 5, 64, (1024, 1024), (256, 128, 64, 32, 16, 8), 9,
                      4, 1000, 500, 0.5, 0.3, num_gt=2000)


def retinanet_anchor_sizes(num_levels: int, first: int = 16):
    """Size tuples the reference builds: x, int(x*2^(1/3)), int(x*2^(2/3)) for x = first*2^i
    (reference object_detection/retinanet/retinanet_model.py:34-42)."""
    return tuple((x, int(x * 2 ** (1.0 / 3)), int(x * 2 ** (2.0 / 3)))
                 for x in [first * 2 ** i for i in range(num_levels)])


def make_gt_boxes(num: int, image_size: Tuple[int, int], gen: torch.Generator) -> torch.Tensor:
    h, w = image_size
    if num == 0:
        return torch.zeros((0, 4), dtype=torch.float32)
    cx = torch.rand(num, generator=gen) * w
    cy = torch.rand(num, generator=gen) * h
    lo, hi = math.log(8.0), math.log(256.0)
    bw = torch.exp(lo + (hi - lo) * torch.rand(num, generator=gen))
    bh = torch.exp(lo + (hi - lo) * torch.rand(num, generator=gen))
    x1 = (cx - bw / 2).clamp(0, w - 2)
    y1 = (cy - bh / 2).clamp(0, h - 2)
    x2 = torch.minimum((cx + bw / 2).clamp(0, w), torch.tensor(float(w)))
    y2 = torch.minimum((cy + bh / 2).clamp(0, h), torch.tensor(float(h)))
    x2 = torch.maximum(x2, x1 + 2)
    y2 = torch.maximum(y2, y1 + 2)
    return torch.stack([x1, y1, x2, y2], dim=1).float()


def make_targets(cfg: PipelineConfig, empty_image: int = -1) -> List[dict]:
    """One target dict per image; image ``empty_image`` gets zero boxes (empty-GT guard)."""
    out = []
    for b in range(cfg.batch):
        gen = torch.Generator().manual_seed(1000 * cfg.cfg_id + b)
        m = 0 if b == empty_image else cfg.num_gt
        boxes = make_gt_boxes(m, cfg.image_size, gen)
        labels = torch.randint(1, max(2, cfg.num_classes), (m,), generator=gen, dtype=torch.int64)
        out.append({"boxes": boxes, "labels": labels})
    return out


# Classification-logit distributions for the synthetic head outputs: (mean, std, fraction saturated at +20).
#   "stress": about half of all logits pass score_thresh = 0.05, far more than any trained detector emits;
#             the parity tests use it because every level then overflows topk_candidates and ties abound.
#   "prior":  centred on the bias the reference's classification head starts from, -log((1 - 0.01) / 0.01)
#             (torchvision RetinaNetClassificationHead prior_probability = 0.01), with enough spread that
#             about 1.4 % of the logits pass 0.05: the three finest levels still overflow topk_candidates.
LOGIT_DISTRIBUTIONS = {
    "stress": (-3.0, 1.5, 1e-3),
    "prior": (-math.log(99.0), 0.75, 1e-4),
}


def make_head_outputs(cfg: PipelineConfig, images: Sequence[int] = None, with_ctrness: bool = False,
                      logits: str = "stress"):
    """cls_logits [B, N, C] and bbox_regression [B, N, 4] (and bbox_ctrness [B, N, 1]) on the CPU."""
    mean, std, hot_frac = LOGIT_DISTRIBUTIONS[logits]
    n, c = cfg.num_anchors, cfg.num_classes
    images = list(range(cfg.batch)) if images is None else list(images)
    cls = torch.empty((len(images), n, c), dtype=torch.float32)
    reg = torch.empty((len(images), n, 4), dtype=torch.float32)
    ctr = torch.empty((len(images), n, 1), dtype=torch.float32) if with_ctrness else None
    for i, b in enumerate(images):
        gen = torch.Generator().manual_seed(1000 * cfg.cfg_id + b + 500)
        cls[i].normal_(mean, std, generator=gen)
        hot = torch.rand((n, c), generator=gen) < hot_frac
        cls[i][hot] = 20.0
        reg[i].normal_(0.0, 0.5, generator=gen)
        big = torch.rand((n, 2), generator=gen) < 1e-4
        reg[i][:, 2:][big] = 5.0
        if with_ctrness:
            ctr[i].normal_(0.0, 1.0, generator=gen)
    out = {"cls_logits": cls, "bbox_regression": reg}
    if with_ctrness:
        out["bbox_ctrness"] = ctr
    return out


def level_sizes(cfg: PipelineConfig) -> List[int]:
    return [cfg.anchors_per_loc * g * g for g in cfg.grids]
