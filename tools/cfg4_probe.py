"""Timing of the candidate selection in the dense regime: 1536x1536, 91 classes, score threshold 0.05.
Usage: cfg4_probe.py [images [topk_candidates]]"""
import dataclasses
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from object_detection_amd import _C, kernels, synthetic  # noqa: E402
from object_detection_amd.retinanet import _image_hw_tensor  # noqa: E402
from object_detection_amd.anchor_utils import AnchorGenerator  # noqa: E402
from tools.microbench import timed  # noqa: E402


def main():
    if os.environ.get("OD_LIB"):                       # experimental build of the library
        _C.LIB_PATH = os.path.abspath(os.environ["OD_LIB"])
    dev = torch.device("cuda:0")
    cfg = dataclasses.replace(synthetic.CFG4, batch=int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    sizes = synthetic.level_sizes(cfg)
    ag = AnchorGenerator(synthetic.retinanet_anchor_sizes(len(cfg.grids)), ((0.5, 1.0, 2.0),) * len(cfg.grids))
    grids = [(g, g) for g in cfg.grids]
    strides = [(cfg.image_size[0] // g, cfg.image_size[1] // g) for g in cfg.grids]
    anchors = kernels.anchor_generate(grids, strides, ag.cell_anchors, dev)
    g = torch.Generator(device=dev).manual_seed(4)
    n = sum(sizes)
    cls = torch.randn((cfg.batch, n, cfg.num_classes), generator=g, device=dev) * 1.5 - 3.0
    reg = torch.randn((cfg.batch, n, 4), generator=g, device=dev) * 0.5
    cls_l, reg_l, anc_l = list(cls.split(sizes, 1)), list(reg.split(sizes, 1)), list(anchors.split(sizes))
    hw = _image_hw_tensor([cfg.image_size] * cfg.batch, dev)
    K = int(sys.argv[2]) if len(sys.argv) > 2 else cfg.topk_candidates
    res = {"images": cfg.batch, "topk_candidates": K}
    for thr in (0.05, 0.5):
        sel = lambda: kernels.select_candidates(cls_l, reg_l, anc_l, hw, thr, K, _C.CODER_BOXCODER, 4.135)
        res[f"thr{thr}"] = timed(sel, reps=3, warm=1)
        cb, cs, cl, _, cc = sel()
        res[f"thr{thr}_nms"] = timed(lambda: kernels.batched_nms_padded(cb, cs, cl, cc, K, True, cfg.nms_thresh, _C.NMS_AUTO_GPU,
                                                                        cfg.detections_per_img, want_keep_idx=False), reps=3, warm=1)
        res[f"thr{thr}_pass_fraction"] = float((torch.sigmoid(cls_l[0][0]) > thr).float().mean().item())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
