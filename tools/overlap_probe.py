"""How do the two halves of a cfg2 step share the GPU?  Throughput (ms per step over 200 steps) of the matching half
alone on 1 / 2 streams, the inference half alone on 1 / 2 / 3 streams, and both together, plus single kernels of the
two halves issued side by side (score filter next to the assign kernel)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd import _C, synthetic  # noqa: E402
from object_detection_amd.retinanet import postprocess_batched  # noqa: E402
from object_detection_amd.retinanet_model import get_model  # noqa: E402
import dataclasses  # noqa: E402

if os.environ.get("OD_LIB"):
    _C.LIB_PATH = os.path.abspath(os.environ["OD_LIB"])
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
anchors_flat, _ = model.anchor_generator._grid_anchors_flat(grids, strides, dev)
anchors_l = list(anchors_flat.split(sizes))
NROT = 4
batches = []
for r in range(NROT):
    c = dataclasses.replace(cfg, cfg_id=cfg.cfg_id + 10 * r)
    head = {k: v.to(dev) for k, v in synthetic.make_head_outputs(c, logits="stress").items()}
    targets = [{k: v.to(dev) for k, v in t.items()} for t in synthetic.make_targets(c, empty_image=c.batch - 1)]
    batches.append(({k: list(v.split(sizes, dim=1)) for k, v in head.items()}, targets))
shapes = [cfg.image_size] * cfg.batch
apl = [anchors_flat] * cfg.batch
sal = [anchors_l] * cfg.batch
streams = []


def post(r):
    sp = batches[r][0]
    return postprocess_batched(sp["cls_logits"], sp["bbox_regression"], sal, shapes, model.score_thresh,
                               model.topk_candidates, model.nms_thresh, model.detections_per_img, _C.CODER_BOXCODER,
                               model.box_coder.bbox_xform_clip, return_padded=True)


from object_detection_amd import kernels  # noqa: E402
from object_detection_amd.retinanet import _image_hw_tensor  # noqa: E402
hw = _image_hw_tensor(shapes, dev)
NO_NMS = False


def post_select_only(r):
    sp = batches[r][0]
    return kernels.select_candidates(sp["cls_logits"], sp["bbox_regression"], anchors_l, hw, model.score_thresh,
                                     model.topk_candidates, _C.CODER_BOXCODER, model.box_coder.bbox_xform_clip,
                                     reuse_outputs=True)


def match(r):
    return model.match_anchors(batches[r][1], apl)


def run(n_post, n_match, steps=200, prio=(0, 0)):
    global streams
    # exactly the streams of this configuration, created together: inference streams first
    streams = [torch.cuda.Stream(dev, priority=prio[0]) for _ in range(n_post)] + [None] * (4 - n_post) + \
              [torch.cuda.Stream(dev, priority=prio[1]) for _ in range(n_match)]

    def step(i):
        r = i % NROT
        if n_post:
            torch.cuda.set_stream(streams[i % n_post])
            (post_select_only if NO_NMS else post)(r)
        if n_match:
            torch.cuda.set_stream(streams[4 + i % n_match])
            match(r)
    for i in range(40):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    return 1e3 * t / steps, 1e3 * t_host / steps


for _ in range(3):
    run(2, 1, 100)
print("library:", _C.LIB_PATH)
for no_nms in (False, True):
    NO_NMS = no_nms
    configs = ((0, 1), (0, 2), (1, 0), (3, 0), (1, 1), (2, 1), (3, 1), (2, 2))
    if os.environ.get("OD_PROBE_CONFIGS"):             # e.g. "3x2,4x2,3x3"
        configs = tuple(tuple(int(v) for v in c.split("x")) for c in os.environ["OD_PROBE_CONFIGS"].split(","))
    for n_post, n_match in configs:
        if no_nms and n_post == 0:
            continue
        ms, host = run(n_post, n_match)
        print(f"{'[no NMS] ' if no_nms else ''}inference streams {n_post}, matching streams {n_match}: {ms:.4f} ms per step "
              f"(host enqueue {host:.4f})")

# in-step duration of every kernel (dispatch time stamps of every 8th launch), inference on 2 streams + matching on 1
NO_NMS = False
_C.profile_stride(8)
print("kernel durations inside the overlapped step (us):")
for k, name in enumerate(_C.KERNEL_NAMES):
    if name not in ("match_init", "match_assign", "match_heavy", "match_restore", "sample", "filter", "filter_redo",
                    "select", "nms"):
        continue
    _C.profile_select(k)
    _C.profile_read(k)
    ms_step, _ = run(2, 1, 200)
    ms, n = _C.profile_read(k)
    _C.profile_select(_C.PROFILE_OFF)
    print(f"  {name:14s} {1e3 * ms / max(n, 1):7.1f}   ({n} launches timed, step {ms_step:.4f} ms)")

# the matching chain without its last kernel (no low-quality restore): does a shorter chain shorten the step?
_C.profile_stride(1)
model.proposal_matcher.allow_low_quality_matches = False
for n_post, n_match in ((0, 1), (0, 2), (2, 1), (3, 1)):
    ms, host = run(n_post, n_match)
    print(f"[no restore kernel] inference streams {n_post}, matching streams {n_match}: {ms:.4f} ms per step")
model.proposal_matcher.allow_low_quality_matches = True
