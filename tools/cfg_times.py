"""Kernel timings of the configurations that are parity-test cases rather than the bench line:
FCOS centre-sampling matching (1024 x 1024, 21 824 locations, 200 GT, 16 images), dense RetinaNet matching
(786 240 anchors, 2000 GT per image, 8 / 16 / 32 images = the per-GPU share of batch 64 on 8 / 4 / 2 GPUs) and the
FCOS postprocess.  Prints one JSON object; pair rates are brute-force equivalents (every anchor-GT pair counts,
although the kernels skip pairs whose boxes cannot overlap), `valu_frac` prices 13 fp32 operations per pair against
the non-FMA VALU peak of 78.6 Tops/s."""
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


def pair_rates(pairs: float, us: float) -> dict:
    return {"pairs_per_s": pairs / (us * 1e-6), "valu_frac_bruteforce_equivalent": 13.0 * pairs / (us * 1e-6) / 78.6e12}


def main():
    if os.environ.get("OD_LIB"):                       # experimental build of the library
        _C.LIB_PATH = os.path.abspath(os.environ["OD_LIB"])
    dev = torch.device("cuda:0")
    res = {}

    # ---- FCOS: centre-sampling matching and postprocess ---------------------------------------------------------
    cfg = synthetic.CFG3
    sizes = synthetic.level_sizes(cfg)
    ag = AnchorGenerator(((8,), (16,), (32,), (64,), (128,)), ((1.0,),) * 5)
    grids = [(g, g) for g in cfg.grids]
    strides = [(cfg.image_size[0] // g, cfg.image_size[1] // g) for g in cfg.grids]
    anchors = kernels.anchor_generate(grids, strides, ag.cell_anchors, dev)
    gts = [t["boxes"].to(dev) for t in synthetic.make_targets(cfg, empty_image=cfg.batch - 1)]
    t = timed(lambda: kernels.match_fcos(anchors, gts, sizes[0], sizes[-1], 1.5))
    us = sum(t.values())
    res["cfg3_fcos_match"] = {"kernels_us": t, "images": cfg.batch, "locations": sum(sizes), "gt_per_image": cfg.num_gt,
                              "algorithmic_MB": cfg.batch * (24.0 * sum(sizes) + 16.0 * cfg.num_gt) / 1e6,
                              **pair_rates(float(sum(g.shape[0] for g in gts)) * sum(sizes), us)}
    head = {k: v.to(dev) for k, v in synthetic.make_head_outputs(cfg, with_ctrness=True).items()}
    head["bbox_regression"] = head["bbox_regression"].abs()
    lv = {k: list(v.split(sizes, 1)) for k, v in head.items()}
    anc_l = list(anchors.split(sizes))
    hw = _image_hw_tensor([cfg.image_size] * cfg.batch, dev)
    K = cfg.topk_candidates

    def fcos_post():
        cb, cs, cl, _, cc = kernels.select_candidates(lv["cls_logits"], lv["bbox_regression"], anc_l, hw, cfg.score_thresh,
                                                      K, _C.CODER_LINEAR, 0.0, ctr_levels=lv["bbox_ctrness"])
        kernels.batched_nms_padded(cb, cs, cl, cc, K, True, cfg.nms_thresh, _C.NMS_AUTO_GPU, cfg.detections_per_img,
                                   want_keep_idx=False)

    res["cfg3_fcos_postprocess"] = {"kernels_us": timed(fcos_post), "images": cfg.batch}

    # ---- dense RetinaNet matching: 2000 GT per image ------------------------------------------------------------------
    cfg5 = synthetic.CFG5
    ag = AnchorGenerator(synthetic.retinanet_anchor_sizes(len(cfg5.grids)), ((0.5, 1.0, 2.0),) * len(cfg5.grids))
    grids = [(g, g) for g in cfg5.grids]
    strides = [(cfg5.image_size[0] // g, cfg5.image_size[1] // g) for g in cfg5.grids]
    anchors = kernels.anchor_generate(grids, strides, ag.cell_anchors, dev)
    n = anchors.shape[0]
    for images in (8, 16, 32):
        part = dataclasses.replace(cfg5, batch=images)
        gts = [t["boxes"].to(dev) for t in synthetic.make_targets(part)]
        t = timed(lambda: kernels.match_retinanet(anchors, gts, cfg5.fg_iou_thresh, cfg5.bg_iou_thresh, True), reps=5, warm=2)
        us = sum(t.values())
        res[f"cfg5_dense_match_{images}_images"] = {
            "kernels_us": t, "us_per_image": us / images, "gt_per_image": cfg5.num_gt, "anchors": n,
            "algorithmic_MB": images * (24.0 * n + 16.0 * cfg5.num_gt) / 1e6,
            "algorithmic_GBps": images * (24.0 * n + 16.0 * cfg5.num_gt) / (us * 1e-6) / 1e9,
            **pair_rates(float(images) * n * cfg5.num_gt, us)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
