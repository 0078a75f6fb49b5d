"""CPU: hand-derived known answers and cross-checks of the two CPU restatements (torch vs scalar C).

The reference holds no fixtures for this path (parity unpinned); these tests pin the checker itself.
"""
import math
import os

import numpy as np
import pytest
import torch

from object_detection_amd import synthetic
from oracle import c_oracle, spec_math, tv_reference as tv

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_base_anchor_kat():
    ag = tv.AnchorGenerator(((32,),), ((0.5, 1.0, 2.0),))
    assert ag.cell_anchors[0].tolist() == [[-23, -11, 23, 11], [-16, -16, 16, 16], [-11, -23, 11, 23]]
    # size tuples the reference builds (retinanet_model.py:42)
    assert synthetic.retinanet_anchor_sizes(6) == ((16, 20, 25), (32, 40, 50), (64, 80, 101), (128, 161, 203),
                                                   (256, 322, 406), (512, 645, 812))


def test_anchor_counts():
    assert synthetic.CFG2.num_anchors == 786240
    assert synthetic.CFG4.num_anchors == 1769040
    assert synthetic.CFG3.num_anchors == 21824
    assert synthetic.CFG1.num_anchors == 49104


def test_grid_anchor_order_matches_c():
    ag = tv.AnchorGenerator(((32, 40, 50),), ((0.5, 1.0, 2.0),))
    a = ag.grid_anchors([(3, 5)], [(8, 16)])[0]
    assert a.shape == (3 * 5 * 9, 4)
    assert np.array_equal(a.numpy(), c_oracle.grid_anchors(ag.cell_anchors[0].numpy(), 3, 5, 8, 16))
    # (y, x, a) order: second location is x = 1
    assert torch.equal(a[9], ag.cell_anchors[0][0] + torch.tensor([16., 0., 16., 0.]))


def test_iou_kat():
    iou = tv.box_iou(torch.tensor([[0., 0., 10., 10.]]), torch.tensor([[5., 5., 15., 15.]]))
    assert iou.view(torch.int32).item() == 0x3E124925          # 25/175 in fp32
    assert c_oracle.box_iou([[0, 0, 10, 10]], [[5, 5, 15, 15]]).view(np.uint32)[0, 0] == 0x3E124925


def test_matcher_kat():
    q = torch.tensor([[.7, .55, .3, .2], [.1, .58, .45, 0.]])
    assert tv.Matcher(.6, .5, False)(q.clone()).tolist() == [0, -2, -1, -1]
    assert tv.Matcher(.6, .5, True)(q.clone()).tolist() == [0, 1, -1, -1]
    assert c_oracle.matcher(q.numpy(), .6, .5, True).tolist() == [0, 1, -1, -1]
    with pytest.raises(ValueError):
        tv.Matcher(.6, .5, True)(torch.zeros((0, 4)))


def test_nms_kat():
    b = torch.tensor([[0., 0., 10., 10.], [1., 1., 11., 11.], [20., 20., 30., 30.]])
    s = torch.tensor([.9, .8, .7])
    assert tv.nms(b, s, .5).tolist() == [0, 2]
    assert c_oracle.nms(b.numpy(), s.numpy(), .5).tolist() == [0, 2]
    z = torch.tensor([[5., 5., 5., 5.], [5., 5., 5., 5.]])       # 0/0 = NaN never suppresses
    assert tv.nms(z, torch.tensor([.5, .4]), .5).tolist() == [0, 1]
    assert c_oracle.nms(z.numpy(), np.float32([.5, .4]), .5).tolist() == [0, 1]
    e = torch.tensor([[0., 0., 10., 10.], [0., 0., 10., 3.]])    # IoU == float(0.3) > double 0.3
    assert tv.nms(e, torch.tensor([.9, .8]), .3).tolist() == [0]
    assert c_oracle.nms(e.numpy(), np.float32([.9, .8]), .3).tolist() == [0]
    t = torch.tensor([[0., 0., 4., 4.], [0., 0., 4., 4.]])       # equal scores: lower index first
    assert tv.nms(t, torch.tensor([.5, .5]), .5).tolist() == [0]


def test_decode_clamp_kat():
    clip = math.log(1000.0 / 16)
    assert abs(clip - 4.1351666) < 1e-6
    a = torch.tensor([[0., 0., 10., 20.]])
    out = tv.BoxCoder().decode_single(torch.tensor([[0., 0., 5.0, 0.]]), a, math="spec")
    w = float(spec_math.od_expf(np.float32(clip))) * 10
    assert out[0, 2] - out[0, 0] == pytest.approx(w, rel=1e-6)
    assert out[0, 3] - out[0, 1] == pytest.approx(20.0)


def test_spec_math_against_libm_and_golden():
    x = np.random.default_rng(1).uniform(-90, 88, 200000).astype(np.float32)
    e = spec_math.od_expf(x)
    ref = np.exp(x.astype(np.float64))
    ok = ref > 1e-37
    assert (np.abs(e[ok] - ref[ok]) / ref[ok]).max() < 2e-7
    assert np.array_equal(e.view(np.uint32), c_oracle.expf(x).view(np.uint32))
    s = spec_math.od_sigmoidf(x / 8)
    st = torch.sigmoid(torch.from_numpy(x / 8)).numpy()
    assert (np.abs(s - st) / st).max() < 1e-6
    g = np.load(os.path.join(GOLD, "spec_math.npz"))
    assert np.array_equal(spec_math.od_expf(g["x"]).view(np.uint32), g["exp"].view(np.uint32))
    assert np.array_equal(spec_math.od_sigmoidf(g["x"]).view(np.uint32), g["sigmoid"].view(np.uint32))
    assert spec_math.od_sigmoidf(np.float32(20.0)) == 1.0 and spec_math.od_sigmoidf(np.float32(-200.0)) == 0.0


@pytest.mark.parametrize("m", [1, 13, 90])
def test_match_restatements_agree(m):
    ag = tv.AnchorGenerator(synthetic.retinanet_anchor_sizes(3, 16), ((0.5, 1.0, 2.0),) * 3)
    anchors = ag.forward((128, 128), [(32, 32), (16, 16), (8, 8)], 1)[0]
    gt = synthetic.make_gt_boxes(m, (128, 128), torch.Generator().manual_seed(m))
    a = tv.retinanet_match([gt], [anchors], 0.6, 0.5)[0]
    assert np.array_equal(a.numpy(), c_oracle.match_retinanet(gt.numpy(), anchors.numpy(), 0.6, 0.5))
    assert torch.equal(a, tv.retinanet_match_chunked(gt, anchors, 0.6, 0.5, chunk=1000))
    assert torch.equal(tv.Matcher(0.6, 0.5, True)(tv.box_iou(gt, anchors)), a)
    assert np.array_equal(c_oracle.matcher(tv.box_iou(gt, anchors).numpy(), 0.6, 0.5), a.numpy())
    fag = tv.AnchorGenerator(((8,), (16,), (32,)), ((1.0,),) * 3)
    fa = fag.forward((128, 128), [(32, 32), (16, 16), (8, 8)], 1)[0]
    f = tv.fcos_match([gt], [fa], [1024, 256, 64], 1.5)[0]
    assert np.array_equal(f.numpy(), c_oracle.match_fcos(gt.numpy(), fa.numpy(), 1024, 64, 1.5))
    # empty targets: all -1
    assert (tv.retinanet_match([torch.zeros((0, 4))], [anchors], 0.6, 0.5)[0] == -1).all()
    assert (c_oracle.match_fcos(np.zeros((0, 4), np.float32), fa.numpy(), 1024, 64) == -1).all()


def test_gt_overlapping_no_anchor_restores_everything():
    ag = tv.AnchorGenerator(((16,),), ((1.0,),))
    anchors = ag.forward((64, 64), [(8, 8)], 1)[0]
    gt = torch.tensor([[10., 10., 40., 40.], [900., 900., 910., 910.]])
    m = tv.retinanet_match([gt], [anchors], 0.6, 0.5)[0]
    assert (m >= 0).all()
    assert np.array_equal(m.numpy(), c_oracle.match_retinanet(gt.numpy(), anchors.numpy(), 0.6, 0.5))


@pytest.mark.parametrize("n,classes", [(1, 1), (200, 3), (1200, 4)])
def test_nms_restatements_agree(n, classes):
    gen = torch.Generator().manual_seed(n)
    xy = torch.rand((n, 2), generator=gen) * 200
    boxes = torch.cat([xy, xy + torch.rand((n, 2), generator=gen) * 60 + 1], 1)
    scores = torch.rand(n, generator=gen)
    scores[::5] = scores[0]
    labels = torch.randint(0, classes, (n,), generator=gen)
    for thr in (0.3, 0.5):
        assert np.array_equal(tv.nms(boxes, scores, thr).numpy(), c_oracle.nms(boxes.numpy(), scores.numpy(), thr))
        assert np.array_equal(tv._batched_nms_coordinate_trick(boxes, scores, labels, thr).numpy(),
                              c_oracle.batched_nms(boxes.numpy(), scores.numpy(), labels.numpy(), thr, 0))
        assert np.array_equal(tv._batched_nms_vanilla(boxes, scores, labels, thr).numpy(),
                              c_oracle.batched_nms(boxes.numpy(), scores.numpy(), labels.numpy(), thr, 1))


def test_filter_topk_and_decode_restatements_agree():
    cfg = synthetic.PipelineConfig("t", 8, 1, (128, 128), (32, 16), 9, 3, 50, 20, 0.5, 0.3)
    head = synthetic.make_head_outputs(cfg)
    logits = head["cls_logits"][0]
    s = tv._sigmoid(logits, "spec").flatten()
    keep = s > 0.5
    vals, idx = tv.topk_stable(s[keep], min(50, int(keep.sum())))
    flat = torch.where(keep)[0][idx]
    cs, ci = c_oracle.filter_topk(logits.numpy(), None, 3, 0.5, 50)
    assert np.array_equal(ci, flat.numpy()) and np.array_equal(cs.view(np.uint32), vals.numpy().view(np.uint32))
    # tie-free data: the adopted tie rule coincides with torch.topk
    u = torch.rand(5000, generator=torch.Generator().manual_seed(3))
    assert torch.equal(tv.topk_stable(u, 100)[1], u.topk(100)[1])
    anchors = torch.rand((64, 4)) * 50
    anchors[:, 2:] += anchors[:, :2] + 1
    rel = torch.randn((64, 4))
    rel[3, 2] = 9.0
    d = tv.BoxCoder().decode_single(rel, anchors, math="spec")
    assert np.array_equal(d.numpy().view(np.uint32), c_oracle.decode_single(rel.numpy(), anchors.numpy(),
                                                                           math.log(1000.0 / 16)).view(np.uint32))
    assert torch.allclose(d, tv.BoxCoder().decode_single(rel, anchors, math="torch"), rtol=1e-5, atol=1e-5)
    assert np.array_equal(tv.BoxLinearCoder(True).decode(rel, anchors).numpy(), c_oracle.decode_linear(rel.numpy(), anchors.numpy()))
    assert np.array_equal(tv.clip_boxes_to_image(d, (40, 30)).numpy(), c_oracle.clip(d.numpy(), 40, 30))


def test_golden_fixtures_match_oracle():
    g = np.load(os.path.join(GOLD, "match_small.npz"))
    anchors, gt0, gt1 = (torch.from_numpy(g[k]) for k in ("anchors", "gt0", "gt1"))
    m = tv.retinanet_match([gt0, gt1], [anchors, anchors], 0.6, 0.5)
    assert np.array_equal(m[0].numpy(), g["matched0"]) and np.array_equal(m[1].numpy(), g["matched1"])
    assert np.array_equal(c_oracle.match_fcos(g["gt0"], g["fcos_anchors"], int(g["npl"][0]), int(g["npl"][-1])), g["fcos0"])
    n = np.load(os.path.join(GOLD, "nms_small.npz"))
    assert np.array_equal(c_oracle.batched_nms(n["boxes"], n["scores"], n["labels"], 0.3, 0), n["keep_trick_03"])
    assert np.array_equal(tv._batched_nms_vanilla(torch.from_numpy(n["boxes"]), torch.from_numpy(n["scores"]),
                                                  torch.from_numpy(n["labels"]), 0.3).numpy(), n["keep_vanilla_03"])
    p = np.load(os.path.join(GOLD, "postprocess_small.npz"))
    lv = p["level_sizes"].tolist()
    split = {"cls_logits": list(torch.from_numpy(p["cls_logits"]).split(lv, dim=1)),
             "bbox_regression": list(torch.from_numpy(p["bbox_regression"]).split(lv, dim=1))}
    anchors = list(torch.from_numpy(p["anchors"]).split(lv))
    dets = tv.retinanet_postprocess(split, [anchors] * 2, [tuple(s) for s in p["shapes"].tolist()], 0.5, 200, 0.3, 60,
                                    math="spec", device_rule="gpu")
    for i, d in enumerate(dets):
        assert np.array_equal(d["boxes"].numpy(), p[f"boxes{i}"]) and np.array_equal(d["labels"].numpy(), p[f"labels{i}"])
    # literal torch math agrees within the tolerance stated for decoded boxes / scores (1e-5 relative)
    dt = tv.retinanet_postprocess(split, [anchors] * 2, [tuple(s) for s in p["shapes"].tolist()], 0.5, 200, 0.3, 60,
                                  math="torch", device_rule="gpu")
    for i, d in enumerate(dt):
        assert d["boxes"].shape == p[f"boxes{i}"].shape
        assert np.allclose(d["boxes"].numpy(), p[f"boxes{i}"], rtol=1e-5, atol=1e-4)
        assert np.allclose(d["scores"].numpy(), p[f"scores{i}"], rtol=1e-5)


def test_predict_tail_restatements_agree():
    """reference predict.py:66-76 for one image: the torch restatement against the scalar C route (fp32 products in
    numpy, orc_nms, truncation), both threshold rules, and the hand-derived IoU == float(0.3) case."""
    import numpy as np
    import torch
    from oracle import c_oracle, tv_reference as tv
    gen = torch.Generator().manual_seed(77)
    xy = torch.rand((300, 2), generator=gen) * 200
    boxes = torch.cat([xy, xy + torch.rand((300, 2), generator=gen) * 80 + 1], dim=1)
    scores = torch.rand(300, generator=gen).sort(descending=True)[0]
    labels = torch.randint(1, 4, (300,), generator=gen)
    for float_rule in (False, True):
        pred = {"boxes": boxes.clone(), "scores": scores, "labels": labels}
        tb, ts, tl, tk = tv.predict_tail(pred, (1920, 1080), (1024, 768), 0.3, float_rule)
        ratio = np.array([1920 / 1024, 1080 / 768] * 2, dtype=np.float32)
        scaled = boxes.numpy() * ratio
        ck = c_oracle.nms(scaled, scores.numpy(), 0.3, float_rule)
        assert np.array_equal(ck, tk.numpy()) and 0 < len(ck) < 300
        assert np.array_equal(np.trunc(scaled[ck]).astype(np.int32), tb.numpy())
        assert np.array_equal(pred["boxes"].numpy(), scaled)
        assert torch.equal(ts, scores[tk]) and torch.equal(tl, labels[tk])
    kat = {"boxes": torch.tensor([[0., 0., 20., 20.], [0., 0., 20., 6.], [40., 40., 61., 61.]]),
           "scores": torch.tensor([0.9, 0.8, 0.7]), "labels": torch.tensor([1, 2, 3])}
    b, _, l, k = tv.predict_tail({k_: v.clone() for k_, v in kat.items()}, (512, 512), (1024, 1024), 0.3, False)
    assert k.tolist() == [0, 2] and b.tolist() == [[0, 0, 10, 10], [20, 20, 30, 30]] and l.tolist() == [1, 3]
    _, _, _, k = tv.predict_tail({k_: v.clone() for k_, v in kat.items()}, (512, 512), (1024, 1024), 0.3, True)
    assert k.tolist() == [0, 1, 2]


def test_predict_tail_golden_fixture_matches_oracle():
    """tests/golden/predict_tail_small.npz (made by tests/golden/make_golden.py from the detections of
    postprocess_small.npz): both restatements of reference predict.py:66-76 reproduce it."""
    import numpy as np
    import torch
    from oracle import c_oracle, tv_reference as tv
    g = np.load(os.path.join(GOLD, "predict_tail_small.npz"))
    p = np.load(os.path.join(GOLD, "postprocess_small.npz"))
    model_wh = tuple(int(v) for v in g["model_wh"])
    for i in range(2):
        wh = tuple(int(v) for v in g["image_wh"][i])
        for rule, float_rule in (("cuda", True), ("cpu", False)):
            pred = {k: torch.from_numpy(p[f"{k}{i}"].copy()) for k in ("boxes", "scores", "labels")}
            tb, ts, tl, tk = tv.predict_tail(pred, wh, model_wh, 0.3, float_rule)
            assert np.array_equal(tk.numpy(), g[f"keep{i}_{rule}"]) and np.array_equal(tb.numpy(), g[f"boxes{i}_{rule}"])
            assert np.array_equal(ts.numpy(), g[f"scores{i}_{rule}"]) and np.array_equal(tl.numpy(), g[f"labels{i}_{rule}"])
            assert np.array_equal(pred["boxes"].numpy(), g[f"scaled{i}"])
            ratio = np.array([wh[0] / model_wh[0], wh[1] / model_wh[1]] * 2, dtype=np.float32)
            ck = c_oracle.nms(p[f"boxes{i}"] * ratio, p[f"scores{i}"], 0.3, float_rule)
            assert np.array_equal(ck, g[f"keep{i}_{rule}"])
        assert 0 < len(g[f"keep{i}_cuda"]) < p[f"boxes{i}"].shape[0]          # the second NMS removes something
