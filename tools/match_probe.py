#!/usr/bin/env python3
"""Matching of the bench workload alone (16 images x 786 240 anchors x 100 GT), for counter passes and quick timings:
   python tools/match_probe.py [launches] [allow_low_quality 0/1]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from object_detection_amd import kernels, synthetic  # noqa: E402
from object_detection_amd.retinanet_model import get_model  # noqa: E402


def main():
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    lq = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
    dev = torch.device("cuda:0")
    cfg = synthetic.CFG2
    targets = synthetic.make_targets(cfg, empty_image=cfg.batch - 1)
    boxes = kernels.pack_boxes([t["boxes"].to(dev) for t in targets])
    L = len(cfg.grids)
    model = get_model(cfg.num_classes, layers=list(range(L - 2)), backbone="pyramid_stub", out_channels=32,
                      min_size=cfg.image_size[0], max_size=cfg.image_size[0]).to(dev).eval()
    grids = [(g, g) for g in cfg.grids]
    strides = [(cfg.image_size[0] // g, cfg.image_size[1] // g) for g in cfg.grids]
    anchors, _ = model.anchor_generator._grid_anchors_flat(grids, strides, dev)    # carries the note on the list's order
    assert anchors.shape[0] == cfg.num_anchors
    for _ in range(3):
        kernels.match_retinanet(anchors, boxes, cfg.fg_iou_thresh, cfg.bg_iou_thresh, lq)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(launches):
        kernels.match_retinanet(anchors, boxes, cfg.fg_iou_thresh, cfg.bg_iou_thresh, lq)
    torch.cuda.synchronize()
    print(f"match lq={int(lq)}: {(time.perf_counter() - t0) / launches * 1e6:.1f} us per call (back to back)")


if __name__ == "__main__":
    main()
