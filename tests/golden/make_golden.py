"""Generates tests/golden/*.npz from the CPU checkers (oracle/), seeded.

SELF-ORACLE FIXTURES, torchvision-unverified: torchvision 0.16.2 is not installable here and the
reference holds no vectors of its own, so these files pin the behaviour of this repository's
restatement (they catch regressions of the checker and of the HIP kernels), not of torchvision.
Run from the repository root:  python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from object_detection_amd import synthetic  # noqa: E402
from oracle import c_oracle, tv_reference as tv  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def small_cfg():
    return synthetic.PipelineConfig("golden_small", 9, 2, (128, 128), (32, 16, 8), 9, 3, 200, 60, 0.5, 0.3, num_gt=12)


def main():
    cfg = small_cfg()
    sizes = synthetic.retinanet_anchor_sizes(3, 16)
    ag = tv.AnchorGenerator(sizes, ((0.5, 1.0, 2.0),) * 3)
    anchors = ag.forward(cfg.image_size, [(g, g) for g in cfg.grids], 1)[0]
    targets = synthetic.make_targets(cfg)
    gt0, gt1 = targets[0]["boxes"], targets[1]["boxes"]
    m0 = tv.retinanet_match([gt0, gt1], [anchors, anchors], 0.6, 0.5)
    assert np.array_equal(m0[0].numpy(), c_oracle.match_retinanet(gt0.numpy(), anchors.numpy(), 0.6, 0.5))
    fag = tv.AnchorGenerator(((8,), (16,), (32,)), ((1.0,),) * 3)
    fanchors = fag.forward(cfg.image_size, [(g, g) for g in cfg.grids], 1)[0]
    npl = [g * g for g in cfg.grids]
    f0 = tv.fcos_match([gt0, gt1], [fanchors, fanchors], npl, 1.5)
    np.savez_compressed(os.path.join(OUT, "match_small.npz"), anchors=anchors.numpy(), gt0=gt0.numpy(), gt1=gt1.numpy(),
                        matched0=m0[0].numpy(), matched1=m0[1].numpy(), fcos_anchors=fanchors.numpy(),
                        fcos0=f0[0].numpy(), fcos1=f0[1].numpy(), npl=np.asarray(npl))

    gen = torch.Generator().manual_seed(77)
    n = 400
    xy = torch.rand((n, 2), generator=gen) * 120
    wh = torch.rand((n, 2), generator=gen) * 40 + 1
    boxes = torch.cat([xy, xy + wh], 1)
    scores = torch.rand(n, generator=gen)
    scores[::9] = scores[3]
    labels = torch.randint(0, 3, (n,), generator=gen)
    np.savez_compressed(os.path.join(OUT, "nms_small.npz"), boxes=boxes.numpy(), scores=scores.numpy(), labels=labels.numpy(),
                        keep_trick_03=c_oracle.batched_nms(boxes.numpy(), scores.numpy(), labels.numpy(), 0.3, 0),
                        keep_vanilla_03=c_oracle.batched_nms(boxes.numpy(), scores.numpy(), labels.numpy(), 0.3, 1),
                        keep_plain_05=c_oracle.nms(boxes.numpy(), scores.numpy(), 0.5))

    head = synthetic.make_head_outputs(cfg)
    lv = synthetic.level_sizes(cfg)
    split = {k: list(v.split(lv, dim=1)) for k, v in head.items()}
    shapes = [(128, 128), (120, 100)]
    dets = tv.retinanet_postprocess(split, [list(anchors.split(lv))] * 2, shapes, cfg.score_thresh, cfg.topk_candidates,
                                    cfg.nms_thresh, cfg.detections_per_img, math="spec", device_rule="gpu",
                                    return_internals=True)
    out = dict(cls_logits=head["cls_logits"].numpy(), bbox_regression=head["bbox_regression"].numpy(),
               anchors=anchors.numpy(), shapes=np.asarray(shapes), level_sizes=np.asarray(lv))
    for i, d in enumerate(dets):
        for k in ("boxes", "scores", "labels", "keep", "cand_boxes", "cand_scores", "cand_labels"):
            out[f"{k}{i}"] = d[k].numpy()
    np.savez_compressed(os.path.join(OUT, "postprocess_small.npz"), **out)

    # tail of the reference's prediction loop (predict.py:66-76) on the detections above: rescale to the size of the
    # original image, class-agnostic NMS at 0.3 under both threshold rules, int32 boxes
    tail = dict(image_wh=np.asarray([(1920, 1080), (100, 161)]), model_wh=np.asarray((128, 128)))
    for i, d in enumerate(dets):
        for rule, float_rule in (("cuda", True), ("cpu", False)):
            pred = {k: d[k].clone() for k in ("boxes", "scores", "labels")}
            tb, ts, tl, tk = tv.predict_tail(pred, tuple(int(v) for v in tail["image_wh"][i]), (128, 128), 0.3, float_rule)
            tail[f"boxes{i}_{rule}"], tail[f"scores{i}_{rule}"] = tb.numpy(), ts.numpy()
            tail[f"labels{i}_{rule}"], tail[f"keep{i}_{rule}"] = tl.numpy(), tk.numpy()
            tail[f"scaled{i}"] = pred["boxes"].numpy()
    np.savez_compressed(os.path.join(OUT, "predict_tail_small.npz"), **tail)

    x = np.concatenate([np.linspace(-104, 89, 4001), [-200.0, 100.0, 0.0, math.log(1000 / 16)]]).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "spec_math.npz"), x=x, exp=c_oracle.expf(x), sigmoid=c_oracle.sigmoidf(x))
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".npz")))


if __name__ == "__main__":
    main()
