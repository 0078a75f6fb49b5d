"""Matching kernel times of the bench workload (or of the same pyramid on a SIZE x SIZE image) with the grid hint
restricted to the first K pyramid levels.
Usage: python tools/match_grids_probe.py [SIZE [lib.so]]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from object_detection_amd import _C, kernels, synthetic  # noqa: E402
from object_detection_amd.anchor_utils import AnchorGenerator  # noqa: E402
from tools.microbench import timed  # noqa: E402


def main():
    if len(sys.argv) > 2:
        _C.LIB_PATH = os.path.abspath(sys.argv[2])
    dev = torch.device("cuda:0")
    cfg = synthetic.CFG2
    if len(sys.argv) > 1 and int(sys.argv[1]) != cfg.image_size[0]:
        import dataclasses
        size = int(sys.argv[1])
        cfg = dataclasses.replace(cfg, image_size=(size, size), grids=tuple(size // st for st in (4, 8, 16, 32, 64, 128)))
    ag = AnchorGenerator(synthetic.retinanet_anchor_sizes(len(cfg.grids)), ((0.5, 1.0, 2.0),) * len(cfg.grids))
    anchors = kernels.anchor_generate([(g, g) for g in cfg.grids], [(cfg.image_size[0] // g,) * 2 for g in cfg.grids],
                                      ag.cell_anchors, dev)
    gts = [t["boxes"].to(dev) for t in synthetic.make_targets(cfg, empty_image=cfg.batch - 1)]
    full = anchors._od_anchor_grids
    res = {"image": cfg.image_size[0], "anchors": int(anchors.shape[0]), "grids": full}
    for k in range(len(full) + 1):
        res[f"levels_{k}"] = timed(lambda: kernels.match_retinanet(anchors, gts, cfg.fg_iou_thresh, cfg.bg_iou_thresh, True,
                                                                   grids=full[:k]))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
