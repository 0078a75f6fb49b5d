"""Host cost of the model's own postprocess_detections (lazy per-image dictionaries) against the padded form of the
same call: burst timings into empty queues and a cProfile of the enqueue path."""
import cProfile
import os
import pstats
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd import _C, synthetic  # noqa: E402
from object_detection_amd.retinanet import postprocess_batched  # noqa: E402
from object_detection_amd.retinanet_model import get_model  # noqa: E402

dev = torch.device("cuda:0")
cfg = synthetic.CFG2
L = len(cfg.grids)
model = get_model(cfg.num_classes, layers=list(range(L - 2)), backbone="pyramid_stub", out_channels=32,
                  score_thresh=cfg.score_thresh, nms_thresh=cfg.nms_thresh, topk_candidates=cfg.topk_candidates,
                  detections_per_img=cfg.detections_per_img, fg_iou_thresh=cfg.fg_iou_thresh,
                  bg_iou_thresh=cfg.bg_iou_thresh, min_size=1024, max_size=1024).to(dev).eval()
sizes = synthetic.level_sizes(cfg)
grids = [(g, g) for g in cfg.grids]
strides = [(cfg.image_size[0] // g, cfg.image_size[1] // g) for g in cfg.grids]
anchors_l = model.anchor_generator.grid_anchors(grids, strides, dev)
head = {k: v.to(dev) for k, v in synthetic.make_head_outputs(cfg).items()}
split = {k: list(v.split(sizes, dim=1)) for k, v in head.items()}
shapes = [cfg.image_size] * cfg.batch
sal = [anchors_l] * cfg.batch
pinned = torch.empty((64, cfg.batch), dtype=torch.int32).pin_memory()
rows = list(pinned.unbind(0))
it = [0]


def padded():
    out = postprocess_batched(split["cls_logits"], split["bbox_regression"], sal, shapes, model.score_thresh,
                              model.topk_candidates, model.nms_thresh, model.detections_per_img, _C.CODER_BOXCODER,
                              model.box_coder.bbox_xform_clip, return_padded=True)
    it[0] += 1
    rows[it[0] % 64].copy_(out[3], non_blocking=True)
    return out


def api():
    return model.postprocess_detections(split, sal, shapes)


def burst(fn, n=4, reps=40):
    samples = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        samples.append((time.perf_counter() - t0) / n * 1e6)
    torch.cuda.synchronize()
    return min(samples), statistics.median(samples)


for name, fn in (("padded", padded), ("api", api)):
    for _ in range(20):
        fn()
    print(name, "host per call, bursts of 4: min %.1f us, median %.1f us" % burst(fn))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    api()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
