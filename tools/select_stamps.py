"""Time stamps of the select kernel (a build with -DOD_SEL_STAMPS writes them over the last 16 index slots of every
(image, level)): tools/select_stamps.py LIB [logits]"""
import dataclasses
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from object_detection_amd import _C, kernels, synthetic  # noqa: E402
from object_detection_amd.retinanet import _image_hw_tensor  # noqa: E402
from object_detection_amd.anchor_utils import AnchorGenerator  # noqa: E402


def main():
    _C.LIB_PATH = os.path.abspath(sys.argv[1])
    logits = sys.argv[2] if len(sys.argv) > 2 else "stress"
    dev = torch.device("cuda:0")
    fcos = logits == "cfg3"
    cfg = synthetic.CFG3 if fcos else synthetic.CFG2
    sizes = synthetic.level_sizes(cfg)
    grids = [(g, g) for g in cfg.grids]
    strides = [(cfg.image_size[0] // g, cfg.image_size[1] // g) for g in cfg.grids]
    if fcos:
        ag = AnchorGenerator(tuple((8 * 2 ** i,) for i in range(len(cfg.grids))), ((1.0,),) * len(cfg.grids))
        head = synthetic.make_head_outputs(cfg, with_ctrness=True)
        head["bbox_regression"] = head["bbox_regression"].abs()
    else:
        ag = AnchorGenerator(synthetic.retinanet_anchor_sizes(len(cfg.grids)), ((0.5, 1.0, 2.0),) * len(cfg.grids))
        head = synthetic.make_head_outputs(cfg, logits=logits)
    anchors = kernels.anchor_generate(grids, strides, ag.cell_anchors, dev)
    lv = {k: list(v.to(dev).split(sizes, 1)) for k, v in head.items()}
    anc_l = list(anchors.split(sizes))
    hw = _image_hw_tensor([cfg.image_size] * cfg.batch, dev)
    K = cfg.topk_candidates
    for _ in range(5):
        if fcos:
            out = kernels.select_candidates(lv["cls_logits"], lv["bbox_regression"], anc_l, hw, cfg.score_thresh, K,
                                            _C.CODER_LINEAR, 0.0, ctr_levels=lv["bbox_ctrness"], want_index=True)
        else:
            out = kernels.select_candidates(lv["cls_logits"], lv["bbox_regression"], anc_l, hw, cfg.score_thresh, K,
                                            _C.CODER_BOXCODER, 4.135, want_index=True)
    torch.cuda.synchronize()
    idx = out[3].view(cfg.batch, len(sizes), K)[:, :, K - 16:].cpu()
    for b in (0, 7):
        for l in range(len(sizes)):
            st = [int(v) for v in idx[b, l, :15].tolist() if v != 0]
            print(f"image {b} level {l} keys {int(idx[b, l, 15])} count {int(out[4][b, l])}: "
                  + " ".join(f"{(v - st[0]) / 100.0:.2f}" for v in st))


main()
