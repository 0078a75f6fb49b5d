"""GPU: the pipeline through the model API (RetinaNet / FCOS objects) against the CPU checker and the
committed golden fixtures; full-size (BASELINE configs[1]) checks by direct comparison on two images
and by size-independent properties on the whole batch."""
import dataclasses
import math
import os

import numpy as np
import pytest
import torch

from object_detection_amd import _C, boxes as box_ops, kernels, synthetic
from object_detection_amd.retinanet import postprocess_batched
from object_detection_amd.retinanet_model import get_model
from oracle import c_oracle, tv_reference as tv

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _model(cfg, device, arch="retinanet", **kw):
    L = len(cfg.grids)
    if arch == "retinanet":
        m = get_model(cfg.num_classes, layers=list(range(L - 2)), backbone="pyramid_stub", out_channels=32,
                      score_thresh=cfg.score_thresh, nms_thresh=cfg.nms_thresh, topk_candidates=cfg.topk_candidates,
                      detections_per_img=cfg.detections_per_img, fg_iou_thresh=cfg.fg_iou_thresh,
                      bg_iou_thresh=cfg.bg_iou_thresh, **kw)
    else:
        m = get_model(cfg.num_classes, arch="fcos", backbone="pyramid_stub", out_channels=32,
                      score_thresh=cfg.score_thresh, nms_thresh=cfg.nms_thresh, topk_candidates=cfg.topk_candidates,
                      detections_per_img=cfg.detections_per_img, **kw)
    return m.to(device).eval()


def _anchors(model, cfg, device):
    grids = [(g, g) for g in cfg.grids]
    strides = [(cfg.image_size[0] // g, cfg.image_size[1] // g) for g in cfg.grids]
    return model.anchor_generator.grid_anchors(grids, strides, device)


def _assert_dets_equal(dets, ref):
    assert len(dets) == len(ref)
    for i, (d, r) in enumerate(zip(dets, ref)):
        assert d["boxes"].shape == r["boxes"].shape, f"image {i}: {d['boxes'].shape} vs {r['boxes'].shape}"
        assert torch.equal(d["labels"].cpu(), r["labels"]), f"image {i} labels"
        assert torch.equal(d["scores"].cpu(), r["scores"]), f"image {i} scores"
        assert torch.equal(d["boxes"].cpu(), r["boxes"]), f"image {i} boxes"


def _c_postprocess_image(cls_levels, reg_levels, anchor_levels, image_shape, score_thresh, K, nms_thresh, D, device_rule):
    """RetinaNet.postprocess_detections of one image with the scalar C checker: per level score filter + top-k, decode,
    clip; then batched_nms (branch and threshold rule of `device_rule`) and keep[:D].  Inputs are numpy arrays."""
    boxes, scores, labels = [], [], []
    clipv = math.log(1000.0 / 16)
    for logits, reg, anc in zip(cls_levels, reg_levels, anchor_levels):
        C = logits.shape[-1]
        s, idx = c_oracle.filter_topk(logits, None, C, score_thresh, K)
        a_idx = idx // C
        bx = c_oracle.clip(c_oracle.decode_single(reg[a_idx], anc[a_idx], clipv), float(image_shape[0]), float(image_shape[1]))
        boxes.append(bx); scores.append(s); labels.append(idx % C)
    boxes, scores, labels = np.concatenate(boxes), np.concatenate(scores), np.concatenate(labels)
    limit = 4000 if device_rule == "cpu" else 20000
    keep = c_oracle.batched_nms(boxes, scores, labels, nms_thresh, 1 if boxes.size > limit else 0,
                                float_rule=(device_rule == "gpu"))[:D]
    return boxes[keep], scores[keep], labels[keep]


def test_golden_fixtures(device):
    g = np.load(os.path.join(GOLD, "match_small.npz"))
    anchors = torch.from_numpy(g["anchors"]).to(device)
    gts = [torch.from_numpy(g["gt0"]).to(device), torch.from_numpy(g["gt1"]).to(device)]
    m = kernels.match_retinanet(anchors, gts, 0.6, 0.5, True).cpu().numpy()
    assert np.array_equal(m[0], g["matched0"]) and np.array_equal(m[1], g["matched1"])
    f = kernels.match_fcos(torch.from_numpy(g["fcos_anchors"]).to(device), gts, int(g["npl"][0]), int(g["npl"][-1]), 1.5)
    assert np.array_equal(f[0].cpu().numpy(), g["fcos0"]) and np.array_equal(f[1].cpu().numpy(), g["fcos1"])
    n = np.load(os.path.join(GOLD, "nms_small.npz"))
    b, s, l = (torch.from_numpy(n[k]).to(device) for k in ("boxes", "scores", "labels"))
    assert np.array_equal(box_ops.batched_nms(b, s, l, 0.3, mode=_C.NMS_TRICK).cpu().numpy(), n["keep_trick_03"])
    assert np.array_equal(box_ops.batched_nms(b, s, l, 0.3, mode=_C.NMS_VANILLA).cpu().numpy(), n["keep_vanilla_03"])
    assert np.array_equal(box_ops.nms(b, s, 0.5).cpu().numpy(), n["keep_plain_05"])
    p = np.load(os.path.join(GOLD, "postprocess_small.npz"))
    lv = p["level_sizes"].tolist()
    cls = torch.from_numpy(p["cls_logits"]).to(device)
    reg = torch.from_numpy(p["bbox_regression"]).to(device)
    anc = list(torch.from_numpy(p["anchors"]).to(device).split(lv))
    dets = postprocess_batched(list(cls.split(lv, dim=1)), list(reg.split(lv, dim=1)), [anc, anc],
                               [tuple(x) for x in p["shapes"].tolist()], 0.5, 200, 0.3, 60, _C.CODER_BOXCODER,
                               math.log(1000.0 / 16))
    for i, d in enumerate(dets):
        assert np.array_equal(d["boxes"].cpu().numpy(), p[f"boxes{i}"])
        assert np.array_equal(d["scores"].cpu().numpy(), p[f"scores{i}"])
        assert np.array_equal(d["labels"].cpu().numpy(), p[f"labels{i}"])
    s = np.load(os.path.join(GOLD, "spec_math.npz"))
    x = torch.from_numpy(s["x"]).to(device)
    # exp through the decode kernel: anchors of width 1 centred at 0 -> x2 - x1 == exp(dw)
    rel = torch.zeros((x.numel(), 4), device=device)
    rel[:, 2] = x
    an = torch.tensor([[-0.5, -0.5, 0.5, 0.5]], device=device).repeat(x.numel(), 1)
    out = kernels.decode_boxes(rel, an, _C.CODER_BOXCODER, 1e30).cpu().numpy()
    e = s["exp"]
    half = np.float32(0.5) * e
    assert np.array_equal(out[:, 2], np.float32(0) + half) and np.array_equal(out[:, 0], np.float32(0) - half)


@pytest.mark.parametrize("rule", ["gpu", "cpu"])
def test_retinanet_postprocess_cfg1(device, rule):
    cfg = synthetic.CFG1
    model = _model(cfg, device)
    anchors = _anchors(model, cfg, device)
    sizes = synthetic.level_sizes(cfg)
    head = synthetic.make_head_outputs(cfg)
    shapes = [(512, 512), (500, 470)]
    split_d = {k: [t.to(device) for t in v.split(sizes, dim=1)] for k, v in head.items()}
    split_c = {k: list(v.split(sizes, dim=1)) for k, v in head.items()}
    if rule == "gpu":
        dets = model.postprocess_detections(split_d, [anchors] * cfg.batch, shapes)
    else:
        dets = postprocess_batched(split_d["cls_logits"], split_d["bbox_regression"], [anchors] * cfg.batch, shapes,
                                   cfg.score_thresh, cfg.topk_candidates, cfg.nms_thresh, cfg.detections_per_img,
                                   _C.CODER_BOXCODER, model.box_coder.bbox_xform_clip, nms_mode=_C.NMS_AUTO_CPU)
    ref = tv.retinanet_postprocess(split_c, [[a.cpu() for a in anchors]] * cfg.batch, shapes, cfg.score_thresh,
                                   cfg.topk_candidates, cfg.nms_thresh, cfg.detections_per_img, math="spec",
                                   device_rule=rule)
    _assert_dets_equal(dets, ref)
    assert all(d["boxes"].shape[0] == cfg.detections_per_img for d in dets)
    # the literal torch-math restatement (torch.sigmoid / torch.exp) agrees within 1e-5 relative
    lit = tv.retinanet_postprocess(split_c, [[a.cpu() for a in anchors]] * cfg.batch, shapes, cfg.score_thresh,
                                   cfg.topk_candidates, cfg.nms_thresh, cfg.detections_per_img, math="torch",
                                   device_rule=rule)
    differing = 0
    for d, r in zip(dets, lit):
        if d["boxes"].shape != r["boxes"].shape or not torch.equal(d["labels"].cpu(), r["labels"]):
            differing += 1                       # a last-bit difference of sigmoid / exp reordered or replaced a detection
            continue
        assert torch.allclose(d["scores"].cpu(), r["scores"], rtol=1e-5, atol=0)
        assert torch.allclose(d["boxes"].cpu(), r["boxes"], rtol=1e-5, atol=1e-4)
    assert differing == 0, f"{differing} of {len(dets)} images select other detections with torch's sigmoid / exp"


def test_fcos_postprocess_and_match(device):
    cfg = dataclasses.replace(synthetic.CFG3, batch=2, image_size=(512, 512), grids=(64, 32, 16, 8, 4), num_gt=40)
    model = _model(cfg, device, arch="fcos")
    anchors = _anchors(model, cfg, device)
    sizes = synthetic.level_sizes(cfg)
    head = synthetic.make_head_outputs(cfg, with_ctrness=True)
    head["bbox_regression"] = head["bbox_regression"].abs()           # FCOS regression passes through relu
    shapes = [(512, 512), (512, 400)]
    split_d = {k: [t.to(device) for t in v.split(sizes, dim=1)] for k, v in head.items()}
    split_c = {k: list(v.split(sizes, dim=1)) for k, v in head.items()}
    dets = model.postprocess_detections(split_d, [anchors] * 2, shapes)
    ref = tv.fcos_postprocess(split_c, [[a.cpu() for a in anchors]] * 2, shapes, cfg.score_thresh, cfg.topk_candidates,
                              cfg.nms_thresh, cfg.detections_per_img, math="spec", device_rule="gpu")
    _assert_dets_equal(dets, ref)
    targets = synthetic.make_targets(cfg, empty_image=1)
    flat = torch.cat(anchors)
    got = model.match_anchors([{k: v.to(device) for k, v in t.items()} for t in targets], [flat] * 2, sizes)
    want = tv.fcos_match([t["boxes"] for t in targets], [flat.cpu()] * 2, sizes, 1.5)
    for g, w in zip(got, want):
        assert torch.equal(g.cpu(), w)


def test_fcos_training_step_end_to_end(device):
    """model(images, targets) of the FCOS model in train mode: the three losses come from the fused kernels, are
    finite, and backpropagate into the head parameters; also under autocast."""
    torch.manual_seed(1)
    model = get_model(4, arch="fcos", backbone="pyramid_stub", out_channels=32, min_size=128, max_size=128).to(device)
    images = [torch.rand(3, 128, 128, device=device), torch.rand(3, 100, 120, device=device)]
    targets = [{"boxes": torch.tensor([[10., 10., 60., 70.], [30., 40., 90., 100.]], device=device),
                "labels": torch.tensor([1, 2], device=device)},
               {"boxes": torch.zeros((0, 4), device=device), "labels": torch.zeros((0,), dtype=torch.int64, device=device)}]
    model.train()
    for amp in (False, True):
        model.zero_grad()
        with torch.autocast("cuda", dtype=torch.float16, enabled=amp):
            losses = model(images, targets)
        assert set(losses) == {"classification", "bbox_regression", "bbox_ctrness"}
        for k, v in losses.items():
            assert v.dtype is torch.float32 and bool(torch.isfinite(v)), k
            assert type(v.grad_fn).__name__.startswith(("_RetinaNetFocalLoss", "_FCOSBoxLoss")), (k, type(v.grad_fn).__name__)
        sum(losses.values()).backward()
        g = model.head.regression_head.bbox_ctrness.weight.grad
        assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().sum()) > 0


def test_fcos_full_size_cfg3(device):
    """The FCOS configuration as specified (16 images, 1024 x 1024, 21 824 locations, 200 GT each): centre-sampling
    matches and detections of every image against the CPU checker, bit for bit."""
    cfg = synthetic.CFG3
    model = _model(cfg, device, arch="fcos")
    anchors = _anchors(model, cfg, device)
    sizes = synthetic.level_sizes(cfg)
    assert sum(sizes) == 21824
    head = synthetic.make_head_outputs(cfg, with_ctrness=True)
    head["bbox_regression"] = head["bbox_regression"].abs()
    shapes = [cfg.image_size] * cfg.batch
    split_d = {k: [t.to(device) for t in v.split(sizes, dim=1)] for k, v in head.items()}
    split_c = {k: list(v.split(sizes, dim=1)) for k, v in head.items()}
    dets = model.postprocess_detections(split_d, [anchors] * cfg.batch, shapes)
    ref = tv.fcos_postprocess(split_c, [[a.cpu() for a in anchors]] * cfg.batch, shapes, cfg.score_thresh,
                              cfg.topk_candidates, cfg.nms_thresh, cfg.detections_per_img, math="spec", device_rule="gpu")
    _assert_dets_equal(dets, ref)
    targets = synthetic.make_targets(cfg, empty_image=cfg.batch - 1)
    flat = torch.cat(anchors)
    got = model.match_anchors([{k: v.to(device) for k, v in t.items()} for t in targets], [flat] * cfg.batch, sizes)
    want = tv.fcos_match([t["boxes"] for t in targets], [flat.cpu()] * cfg.batch, sizes, 1.5)
    for g, w in zip(got, want):
        assert torch.equal(g.cpu(), w)


def test_forward_end_to_end_shapes(device):
    """model(images) / model(images, targets) through transform, stub backbone, head and the kernels."""
    torch.manual_seed(0)
    model = get_model(4, backbone="pyramid_stub", layers=[0], out_channels=32, min_size=128, max_size=128,
                      score_thresh=0.4, nms_thresh=0.3, detections_per_img=50).to(device)
    images = [torch.rand(3, 128, 128, device=device), torch.rand(3, 100, 120, device=device)]
    targets = [{"boxes": torch.tensor([[10., 10., 60., 70.], [30., 40., 90., 100.]], device=device),
                "labels": torch.tensor([1, 2], device=device)},
               {"boxes": torch.zeros((0, 4), device=device), "labels": torch.zeros((0,), dtype=torch.int64, device=device)}]
    model.train()
    losses = model(images, targets)
    assert set(losses) == {"classification", "bbox_regression"}
    total = sum(losses.values())
    total.backward()
    assert torch.isfinite(total)
    model.eval()
    with torch.no_grad():
        dets = model(images)
    assert len(dets) == 2
    for d in dets:
        assert d["boxes"].shape[1] == 4 and d["boxes"].shape[0] == d["scores"].shape[0] == d["labels"].shape[0] <= 50
        assert d["labels"].dtype == torch.int64
        assert (d["scores"][:-1] >= d["scores"][1:]).all()


@pytest.mark.parametrize("arch", ["retinanet", "fcos"])
def test_eval_forward_reads_head_maps_in_place(device, arch):
    """model(images) in eval mode takes the channel-major route (no permuted copies of the head outputs) and returns
    exactly the detections of the torchvision-shaped route: head.forward -> split -> postprocess_detections."""
    torch.manual_seed(3)
    kw = dict(backbone="pyramid_stub", out_channels=32, min_size=160, max_size=160, score_thresh=0.005, nms_thresh=0.4,
              detections_per_img=60)
    model = (get_model(4, layers=[0, 1], **kw) if arch == "retinanet" else get_model(4, arch="fcos", **kw)).to(device).eval()
    images = [torch.rand(3, 160, 160, device=device), torch.rand(3, 120, 150, device=device)]
    with torch.no_grad():
        dets = model(images)
        # the same through the public pieces, as torchvision's forward strings them together
        sizes = [(int(i.shape[-2]), int(i.shape[-1])) for i in images]
        il, _ = model.transform(images, None)
        feats = list(model.backbone(il.tensors).values())
        out = model.head(feats)
        anchors = model.anchor_generator(il, feats)
        per_level = [f.shape[2] * f.shape[3] * (out["cls_logits"].shape[1] // sum(x.shape[2] * x.shape[3] for x in feats))
                     for f in feats]
        split = {k: list(v.split(per_level, dim=1)) for k, v in out.items()}
        ref = model.postprocess_detections(split, [list(a.split(per_level)) for a in anchors], il.image_sizes)
        ref = model.transform.postprocess(ref, il.image_sizes, sizes)
    assert sum(d["boxes"].shape[0] for d in dets) > 0
    for d, r in zip(dets, ref):
        for k in ("boxes", "scores", "labels"):
            assert torch.equal(d[k], r[k]), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_fcos_fused_head_losses(device, dtype):
    """FCOSHead.compute_loss on the GPU goes through the fused kernels (classification normalised by the batch's
    foreground count, GIoU of the linearly decoded boxes, centre-ness BCE) and agrees with the torch restatement
    evaluated in fp32 on the CPU from the same values: losses 1e-5 relative, gradients 1e-5 of their largest entry
    plus 1e-4 relative per element (plus one rounding of the element type for 16-bit gradients)."""
    cfg = dataclasses.replace(synthetic.CFG3, batch=3, image_size=(512, 512), grids=(64, 32, 16, 8, 4), num_gt=40,
                              cfg_id=31)
    model = _model(cfg, device, arch="fcos")
    anchors_dev = torch.cat(_anchors(model, cfg, device))
    anchors = anchors_dev.cpu()
    sizes = synthetic.level_sizes(cfg)
    head = synthetic.make_head_outputs(cfg, with_ctrness=True)
    head["bbox_regression"] = head["bbox_regression"].abs()
    head = {k: v.to(dtype) for k, v in head.items()}
    targets = synthetic.make_targets(cfg, empty_image=2)
    matched = tv.fcos_match([t["boxes"] for t in targets], [anchors] * cfg.batch, sizes, 1.5)
    assert int(sum((m >= 0).sum() for m in matched)) > 0
    ref_in = {k: v.float().clone().requires_grad_(True) for k, v in head.items()}
    ref = tv.fcos_head_loss(targets, ref_in, [anchors] * cfg.batch, matched)
    (ref["classification"] + 2.0 * ref["bbox_regression"] + 3.0 * ref["bbox_ctrness"]).backward()
    dev_in = {k: v.to(device).requires_grad_(True) for k, v in head.items()}
    t_dev = [{k: v.to(device) for k, v in t.items()} for t in targets]
    got = model.head.compute_loss(t_dev, dev_in, [anchors_dev] * cfg.batch, [m.to(device) for m in matched])
    for k in ref:
        assert got[k].grad_fn is not None and type(got[k].grad_fn).__name__.startswith(("_RetinaNetFocalLoss", "_FCOSBoxLoss")), k
        assert abs(got[k].item() - ref[k].item()) <= 1e-5 * abs(ref[k].item()), (k, got[k].item(), ref[k].item())
    (got["classification"] + 2.0 * got["bbox_regression"] + 3.0 * got["bbox_ctrness"]).backward()
    ulp = float(torch.finfo(dtype).eps) if dtype is not torch.float32 else 0.0
    spacing = 2.0 ** -24 if dtype is torch.float16 else 0.0
    for k in ref_in:
        g, w = dev_in[k].grad, ref_in[k].grad
        assert g.dtype is dtype and g.shape == w.shape
        g = g.float().cpu()
        tol = 1e-5 * w.abs().max() + (1e-4 + ulp) * w.abs() + spacing
        assert bool(((g - w).abs() <= tol).all()), (k, float((g - w).abs().max()))


def test_training_step_under_autocast(device):
    """The reference trains under torch.cuda.amp.autocast (use_amp): the heads emit fp16, anchors are cast to fp16 and
    the loss dictionary is fp32.  The step runs through the fused kernels and stays close to the fp32 step (the
    16-bit rounding of the head outputs and anchors is the only difference)."""
    torch.manual_seed(0)
    model = get_model(4, backbone="pyramid_stub", layers=[0], out_channels=32, min_size=128, max_size=128).to(device)
    images = [torch.rand(3, 128, 128, device=device), torch.rand(3, 100, 120, device=device)]
    targets = [{"boxes": torch.tensor([[10., 10., 60., 70.], [30., 40., 90., 100.]], device=device),
                "labels": torch.tensor([1, 2], device=device)},
               {"boxes": torch.tensor([[5., 8., 50., 40.]], device=device), "labels": torch.tensor([3], device=device)}]
    model.train()
    full = model(images, targets)
    with torch.autocast("cuda", dtype=torch.float16):
        amp = model(images, targets)
        total = sum(amp.values())
    for k in full:
        assert amp[k].dtype is torch.float32 and type(amp[k].grad_fn).__name__.startswith("_RetinaNet"), k
        assert abs(amp[k].item() - full[k].item()) <= 2e-2 * abs(full[k].item()) + 1e-4, (k, amp[k].item(), full[k].item())
    scaler = torch.amp.GradScaler("cuda")
    scaler.scale(total).backward()
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    assert len(grads) > 0 and all(bool(torch.isfinite(g).all()) for g in grads)


def test_compute_loss_matches_checker(device):
    cfg = dataclasses.replace(synthetic.CFG1, num_gt=30)
    model = _model(cfg, device)
    anchors = torch.cat(_anchors(model, cfg, device))
    head = synthetic.make_head_outputs(cfg)
    targets = synthetic.make_targets(cfg, empty_image=1)
    t_d = [{k: v.to(device) for k, v in t.items()} for t in targets]
    h_d = {k: v.to(device) for k, v in head.items()}
    losses = model.compute_loss(t_d, h_d, [anchors] * cfg.batch)
    matched = tv.retinanet_match([t["boxes"] for t in targets], [anchors.cpu()] * cfg.batch, 0.6, 0.5)
    ref = tv.retinanet_head_loss(targets, head, [anchors.cpu()] * cfg.batch, matched, "giou")
    for k in ("classification", "bbox_regression"):
        assert torch.allclose(losses[k].cpu(), ref[k], rtol=1e-4), k


def test_full_size_cfg2_two_images_and_batch_properties(device):
    cfg = synthetic.CFG2
    model = _model(cfg, device)
    anchors = _anchors(model, cfg, device)
    flat = torch.cat(anchors)
    assert flat.shape[0] == 786240
    sizes = synthetic.level_sizes(cfg)
    head = synthetic.make_head_outputs(cfg)
    targets = synthetic.make_targets(cfg, empty_image=cfg.batch - 1)
    shapes = [cfg.image_size] * cfg.batch
    split_d = {k: [t.to(device) for t in v.split(sizes, dim=1)] for k, v in head.items()}
    dets = model.postprocess_detections(split_d, [anchors] * cfg.batch, shapes)
    # (1) every image against the scalar C checker, bit for bit; images 0 and 7 also against the torch restatement
    anc_np = [a.cpu().numpy() for a in anchors]
    cls_np = [t.numpy() for t in head["cls_logits"].split(sizes, dim=1)]
    reg_np = [t.numpy() for t in head["bbox_regression"].split(sizes, dim=1)]
    for i in range(cfg.batch):
        bx, sc, lb = _c_postprocess_image([t[i] for t in cls_np], [t[i] for t in reg_np], anc_np, cfg.image_size,
                                          cfg.score_thresh, cfg.topk_candidates, cfg.nms_thresh, cfg.detections_per_img, "gpu")
        assert np.array_equal(dets[i]["boxes"].cpu().numpy().view(np.uint32), bx.view(np.uint32)), f"image {i} boxes"
        assert np.array_equal(dets[i]["scores"].cpu().numpy().view(np.uint32), sc.view(np.uint32)), f"image {i} scores"
        assert np.array_equal(dets[i]["labels"].cpu().numpy(), lb), f"image {i} labels"
    flips = 0
    for i in (0, 7):
        one = {k: [t[i:i + 1] for t in v.split(sizes, dim=1)] for k, v in head.items()}
        ref = tv.retinanet_postprocess(one, [[a.cpu() for a in anchors]], [cfg.image_size], cfg.score_thresh,
                                       cfg.topk_candidates, cfg.nms_thresh, cfg.detections_per_img, math="spec",
                                       device_rule="gpu")
        _assert_dets_equal([dets[i]], ref)
        # the same image with torch's own sigmoid / exp: how often does the last bit change which boxes come out?
        lit = tv.retinanet_postprocess(one, [[a.cpu() for a in anchors]], [cfg.image_size], cfg.score_thresh,
                                       cfg.topk_candidates, cfg.nms_thresh, cfg.detections_per_img, math="torch",
                                       device_rule="gpu")[0]
        if lit["labels"].shape != ref[0]["labels"].shape or not torch.equal(lit["labels"], ref[0]["labels"]):
            flips += 1
        else:
            assert torch.allclose(ref[0]["scores"], lit["scores"], rtol=1e-5, atol=0)
            assert torch.allclose(ref[0]["boxes"], lit["boxes"], rtol=1e-5, atol=1e-4)
    print(f"cfg2: {flips} of 2 images select other detections with torch's sigmoid / exp than with the specified ones")
    assert flips <= 1
    # (2) batch-size independence: an image alone gives the same detections as inside the batch
    alone = model.postprocess_detections({k: [t[3:4] for t in v] for k, v in split_d.items()}, [anchors], [cfg.image_size])
    _assert_dets_equal(alone, [{k: v.cpu() for k, v in dets[3].items()}])
    # (3) properties on every image: capped, sorted by score, inside the image, NMS-idempotent per class
    for d in dets:
        n = d["boxes"].shape[0]
        assert n <= cfg.detections_per_img
        assert (d["scores"][:-1] >= d["scores"][1:]).all() and (d["scores"] > cfg.score_thresh).all()
        assert (d["boxes"] >= 0).all() and (d["boxes"] <= 1024).all()
        again = box_ops.batched_nms(d["boxes"], d["scores"], d["labels"], cfg.nms_thresh, mode=_C.NMS_VANILLA)
        assert again.numel() == n and torch.equal(again.cpu(), torch.arange(n))
    # (4) matching: bit-exact against the scalar C checker on two images, invariants on all
    t_d = [{k: v.to(device) for k, v in t.items()} for t in targets]
    matched = model.match_anchors(t_d, [flat] * cfg.batch)
    fa = flat.cpu().numpy()
    for i in (0, 5):
        assert np.array_equal(matched[i].cpu().numpy(), c_oracle.match_retinanet(targets[i]["boxes"].numpy(), fa, 0.6, 0.5))
    assert (matched[cfg.batch - 1] == -1).all()                     # empty-target image
    for i in range(cfg.batch - 1):
        m = matched[i]
        assert int(m.max()) < cfg.num_gt and int(m.min()) >= -2
        # the low-quality restore guarantees foreground anchors exist whenever GT boxes overlap anchors
        assert int((m >= 0).sum()) >= 1


def test_full_size_cfg4_selection_properties(device):
    """One image of the largest configuration (1536 x 1536, 1 769 040 anchors, 91 classes, score threshold 0.05: 161 M
    logits, about half of them above the threshold).  The candidate selection is checked through properties that
    hold at any size, evaluated with torch on the GPU, and then bit for bit against the scalar C checker: per level the candidates are
    sorted (score descending, flat index ascending on ties), their scores are the sigmoid of the logits they point
    at (1e-6), there are min(K, #passing) of them, and no logit outside the selection scores higher than the last
    one selected.  The detections are capped, sorted, inside the image and stable under a second NMS."""
    cfg = dataclasses.replace(synthetic.CFG4, batch=1)
    model = _model(cfg, device)
    anchors = _anchors(model, cfg, device)
    sizes = synthetic.level_sizes(cfg)
    assert sum(sizes) == 1769040
    g = torch.Generator(device=device).manual_seed(44)
    hw = torch.tensor([[1536, 1536]], dtype=torch.int32, device=device)
    K = cfg.topk_candidates
    cls_l = [torch.randn((1, n, cfg.num_classes), generator=g, device=device) * 1.5 - 3.0 for n in sizes]
    reg_l = [torch.randn((1, n, 4), generator=g, device=device) * 0.5 for n in sizes]
    cb, cs, cl, ci, cc = kernels.select_candidates(cls_l, reg_l, anchors, hw, cfg.score_thresh, K, _C.CODER_BOXCODER,
                                                   model.box_coder.bbox_xform_clip, want_index=True)
    for lvl, logits in enumerate(cls_l):
        flat = logits.reshape(-1)
        n_pass = int((torch.sigmoid(flat) > cfg.score_thresh).sum())
        count = int(cc[0, lvl])
        assert abs(count - min(K, n_pass)) <= 2, (lvl, count, n_pass)       # torch's sigmoid may differ in the last bit at the threshold
        sc = cs[0, lvl * K:lvl * K + count]
        idx = ci[0, lvl * K:lvl * K + count]
        assert int(idx.min()) >= 0 and int(idx.max()) < flat.numel() and idx.unique().numel() == count
        assert bool((sc[:-1] >= sc[1:]).all())
        ties = sc[:-1] == sc[1:]
        assert bool((idx[:-1][ties] < idx[1:][ties]).all())
        assert torch.allclose(sc, torch.sigmoid(flat[idx]), rtol=1e-6, atol=1e-7)
        assert torch.equal(cl[0, lvl * K:lvl * K + count].long(), idx % cfg.num_classes)
        if count == K:                                                        # nothing better was left out
            rest = flat.clone()
            rest[idx] = -float("inf")
            assert float(torch.sigmoid(rest.max())) <= float(sc[-1]) * (1 + 1e-6)
    # per level: indices and score bits against the scalar C checker (it keeps the K best in a heap: 161 M logits)
    for lvl, logits in enumerate(cls_l):
        s_ref, i_ref = c_oracle.filter_topk(logits[0].cpu().numpy(), None, cfg.num_classes, cfg.score_thresh, K)
        count = int(cc[0, lvl])
        assert count == len(i_ref), (lvl, count, len(i_ref))
        assert np.array_equal(ci[0, lvl * K:lvl * K + count].cpu().numpy(), i_ref), lvl
        assert np.array_equal(cs[0, lvl * K:lvl * K + count].cpu().numpy().view(np.uint32), s_ref.view(np.uint32)), lvl
    split = {"cls_logits": cls_l, "bbox_regression": reg_l}
    det = model.postprocess_detections(split, [anchors], [cfg.image_size])[0]
    bx, sc, lb = _c_postprocess_image([t[0].cpu().numpy() for t in cls_l], [t[0].cpu().numpy() for t in reg_l],
                                      [a.cpu().numpy() for a in anchors], cfg.image_size, cfg.score_thresh, K,
                                      cfg.nms_thresh, cfg.detections_per_img, "gpu")
    assert np.array_equal(det["boxes"].cpu().numpy().view(np.uint32), bx.view(np.uint32))
    assert np.array_equal(det["scores"].cpu().numpy().view(np.uint32), sc.view(np.uint32))
    assert np.array_equal(det["labels"].cpu().numpy(), lb)
    n = det["boxes"].shape[0]
    assert 0 < n <= cfg.detections_per_img
    assert bool((det["scores"][:-1] >= det["scores"][1:]).all()) and bool((det["scores"] > cfg.score_thresh).all())
    assert bool((det["boxes"] >= 0).all()) and bool((det["boxes"] <= 1536).all())
    again = box_ops.batched_nms(det["boxes"], det["scores"], det["labels"], cfg.nms_thresh, mode=_C.NMS_VANILLA)
    assert again.numel() == n and torch.equal(again.cpu(), torch.arange(n))


def test_full_size_cfg5_dense_matching(device):
    """The dense-target configuration at full size (786 240 anchors x 2000 GT per image, 1.57 G pairs): too large for
    the CPU checkers inside a test, so two images are checked bit for bit against the checker's own formulas
    (tv.box_iou, Matcher rules) evaluated chunk by chunk with torch on the GPU -- separate fp32 torch kernels round
    exactly like the CPU ops -- and every image through invariants."""
    cfg = dataclasses.replace(synthetic.CFG5, batch=4)
    model = _model(cfg, device)
    flat = torch.cat(_anchors(model, cfg, device))
    targets = synthetic.make_targets(cfg, empty_image=3)
    gts = [t["boxes"].to(device) for t in targets]
    assert gts[0].shape[0] == 2000 and flat.shape[0] == 786240
    matched = kernels.match_retinanet(flat, gts, cfg.fg_iou_thresh, cfg.bg_iou_thresh, True)
    n = flat.shape[0]
    for i in (0, 2):
        gt = gts[i]
        vals = torch.full((n,), -1.0, device=device)
        arg = torch.zeros((n,), dtype=torch.int64, device=device)
        gmax = torch.empty((gt.shape[0],), device=device)
        for g0 in range(0, gt.shape[0], 50):
            iou = tv.box_iou(gt[g0:g0 + 50], flat)                      # [50, N]
            v, a = iou.max(dim=0)                                       # first maximal index inside the chunk
            better = v > vals                                           # strict: earlier chunks win ties
            arg = torch.where(better, a + g0, arg)
            vals = torch.where(better, v, vals)
            gmax[g0:g0 + 50] = iou.max(dim=1)[0]
        want = arg.clone()
        want[vals < cfg.bg_iou_thresh] = -1
        want[(vals >= cfg.bg_iou_thresh) & (vals < cfg.fg_iou_thresh)] = -2
        restore = torch.zeros((n,), dtype=torch.bool, device=device)
        for g0 in range(0, gt.shape[0], 50):
            restore |= (tv.box_iou(gt[g0:g0 + 50], flat) == gmax[g0:g0 + 50, None]).any(dim=0)
        want[restore] = arg[restore]
        assert torch.equal(matched[i], want), (i, int((matched[i] != want).sum()))
    assert bool((matched[3] == -1).all())
    for i in range(3):
        assert int(matched[i].max()) < 2000 and int(matched[i].min()) >= -2 and int((matched[i] >= 0).sum()) > 0


def test_concurrent_streams_do_not_share_scratch(device):
    """Steps enqueued on different HIP streams may overlap on the GPU; every stream has its own scratch buffers,
    so the results must equal those of the same calls made one after the other."""
    base = synthetic.CFG1
    cfgs = [dataclasses.replace(base, cfg_id=40 + i, num_gt=60) for i in range(3)]
    model = _model(base, device)
    anchors_l = _anchors(model, base, device)
    anchors = torch.cat(anchors_l)
    sizes = synthetic.level_sizes(base)
    shapes = [base.image_size] * base.batch
    work = []
    for cfg in cfgs:
        head = {k: v.to(device) for k, v in synthetic.make_head_outputs(cfg).items()}
        split = {k: list(v.split(sizes, dim=1)) for k, v in head.items()}
        gts = [t["boxes"].to(device) for t in synthetic.make_targets(cfg)]
        work.append((split, gts))

    def run(split, gts):
        out = postprocess_batched(split["cls_logits"], split["bbox_regression"], [anchors_l] * base.batch, shapes,
                                  model.score_thresh, model.topk_candidates, model.nms_thresh, model.detections_per_img,
                                  _C.CODER_BOXCODER, model.box_coder.bbox_xform_clip, return_padded=True)
        return out + (kernels.match_retinanet(anchors, gts, 0.6, 0.5, True),)

    ref = [run(*w) for w in work]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device) for _ in work]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream(device))
    got = [None] * len(work)
    for rep in range(6):                         # several rounds, so that launches of different streams interleave
        for i, w in enumerate(work):
            with torch.cuda.stream(streams[i]):
                got[i] = run(*w)
    torch.cuda.synchronize()
    for i in range(len(work)):
        for a, b in zip(got[i], ref[i]):
            assert torch.equal(a, b), f"stream {i}"


def test_graph_capture_replays_step(device):
    """A whole step (candidate selection + NMS on the capture stream, anchor matching forked onto a second stream) is
    recorded into a HIP graph once and replayed: the replay's outputs equal those of the same calls issued one by one,
    also after new values were copied into the captured input tensors."""
    from object_detection_amd import graph_step
    base = synthetic.CFG1
    model = _model(base, device)
    anchors_l = _anchors(model, base, device)
    anchors = torch.cat(anchors_l)
    sizes = synthetic.level_sizes(base)
    shapes = [base.image_size] * base.batch
    cfgs = [dataclasses.replace(base, cfg_id=60 + i, num_gt=50) for i in range(2)]
    heads = [{k: v.to(device) for k, v in synthetic.make_head_outputs(c).items()} for c in cfgs]
    gts = [[t["boxes"].to(device) for t in synthetic.make_targets(c)] for c in cfgs]
    static_head = {k: v.clone() for k, v in heads[0].items()}
    static_gt = [g.clone() for g in gts[0]]
    split = {k: list(v.split(sizes, dim=1)) for k, v in static_head.items()}
    side = torch.cuda.Stream(device)

    def step():
        main = torch.cuda.current_stream(device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            matched = kernels.match_retinanet(anchors, static_gt, 0.6, 0.5, True)
        out = postprocess_batched(split["cls_logits"], split["bbox_regression"], [anchors_l] * base.batch, shapes,
                                  model.score_thresh, model.topk_candidates, model.nms_thresh, model.detections_per_img,
                                  _C.CODER_BOXCODER, model.box_coder.bbox_xform_clip, return_padded=True)
        main.wait_stream(side)
        return out + (matched,)

    def eager(head, gt):
        sp = {k: list(v.split(sizes, dim=1)) for k, v in head.items()}
        out = postprocess_batched(sp["cls_logits"], sp["bbox_regression"], [anchors_l] * base.batch, shapes,
                                  model.score_thresh, model.topk_candidates, model.nms_thresh, model.detections_per_img,
                                  _C.CODER_BOXCODER, model.box_coder.bbox_xform_clip, return_padded=True)
        return out + (kernels.match_retinanet(anchors, gt, 0.6, 0.5, True),)

    want = [[t.clone() for t in eager(heads[i], gts[i])] for i in range(2)]
    torch.cuda.synchronize()
    captured = graph_step.capture(step, device, side_streams=[side])
    for rep in range(2):
        for i in (0, 1, 0):
            for k in static_head:
                static_head[k].copy_(heads[i][k])
            for dst, src in zip(static_gt, gts[i]):
                dst.copy_(src)
            got = captured.replay()
            torch.cuda.synchronize()
            for a, b, name in zip(got, want[i], ("boxes", "scores", "labels", "counts", "matched")):
                assert torch.equal(a, b), (rep, i, name)
    assert int(want[0][3].min()) > 0 and not torch.equal(want[0][4], want[1][4])
    # The launchers' caches move on behind the graph's back: the table cache is cleared (a loop whose GT counts change
    # every step fills it), the scratch buffers of the capture streams are replaced by larger ones, and the freed
    # memory is handed to somebody else.  The graph holds its own references and replays the same results.
    assert len(captured.held) > 0
    for st in (captured.capture_stream, side):
        with torch.cuda.stream(st):
            for j in range(4200):
                kernels._device_ints((j, j + 1, 7), device)
            for tag in ("match", "select", "nms"):
                kernels._workspace(device, tag, 1 << 28)
    assert len(kernels._small_cache) < 4200                        # the cache was cleared on the way
    from object_detection_amd import retinanet as rn
    rn._hw_tables.clear()
    kernels._candidate_buffers.clear()
    torch.cuda.synchronize()
    trash = [torch.full((1 << 22,), -1, dtype=torch.int32, device=device) for _ in range(8)]   # reuses freed blocks
    for i in (1, 0):
        for k in static_head:
            static_head[k].copy_(heads[i][k])
        for dst, src in zip(static_gt, gts[i]):
            dst.copy_(src)
        got = captured.replay()
        torch.cuda.synchronize()
        for a, b, name in zip(got, want[i], ("boxes", "scores", "labels", "counts", "matched")):
            assert torch.equal(a, b), ("after the caches moved on", i, name)
    del trash
    for st in (captured.capture_stream, side):                     # give the 256 MiB scratch buffers back
        for tag in ("match", "select", "nms"):
            kernels._workspaces.pop((device.index, st.cuda_stream, tag), None)
    torch.cuda.empty_cache()


def test_detections_are_cut_to_size_on_first_access(device):
    """postprocess_detections returns dictionaries whose tensors are cut out of the padded batch outputs when they
    are first looked at (the call itself does not wait for the GPU); they behave like plain dictionaries."""
    from object_detection_amd.retinanet import Detections
    cfg = synthetic.CFG1
    model = _model(cfg, device)
    anchors = _anchors(model, cfg, device)
    sizes = synthetic.level_sizes(cfg)
    head = synthetic.make_head_outputs(cfg)
    shapes = [(512, 512), (500, 470)]
    split = {k: [t.to(device) for t in v.split(sizes, dim=1)] for k, v in head.items()}
    dets = model.postprocess_detections(split, [anchors] * cfg.batch, shapes)
    ob, os_, ol, oc = postprocess_batched(split["cls_logits"], split["bbox_regression"], [anchors] * cfg.batch, shapes,
                                          model.score_thresh, model.topk_candidates, model.nms_thresh,
                                          model.detections_per_img, _C.CODER_BOXCODER, model.box_coder.bbox_xform_clip,
                                          return_padded=True)
    assert all(isinstance(d, Detections) and isinstance(d, dict) and d._pending is not None for d in dets)
    counts = oc.tolist()
    assert set(dets[0].keys()) == {"boxes", "scores", "labels"} and dets[0]._pending is None
    plain = dict(dets[1])                                    # the generic mapping protocol materialises as well
    assert sorted(plain) == ["boxes", "labels", "scores"] and len(dets[1]) == 3 and "boxes" in dets[1]
    for i, d in enumerate(dets):
        assert torch.equal(d["boxes"], ob[i, :counts[i]]) and torch.equal(d["scores"], os_[i, :counts[i]])
        assert torch.equal(d["labels"], ol[i, :counts[i]]) and d["labels"].dtype is torch.int64
    # the rescale of transform.postprocess is applied when the boxes are cut out
    again = model.postprocess_detections(split, [anchors] * cfg.batch, shapes)
    model.transform.eval()
    scaled = model.transform.postprocess(again, shapes, [(1024, 1024), (250, 940)])
    assert scaled[0]._pending is not None
    ref = tv.GeneralizedRCNNTransform.postprocess([{"boxes": ob[i, :counts[i]].cpu()} for i in range(cfg.batch)], shapes,
                                                  [(1024, 1024), (250, 940)])
    for d, r in zip(scaled, ref):
        assert torch.equal(d["boxes"].cpu(), r["boxes"])
    scaled[0]["boxes"] = scaled[0]["boxes"] + 1.0            # item assignment works on the materialised dictionary
    assert torch.equal(scaled[0]["boxes"].cpu(), ref[0]["boxes"] + 1.0)
    # the rest of the mapping protocol on dictionaries nobody has looked into yet
    fresh = model.postprocess_detections(split, [anchors] * cfg.batch, shapes)
    assert fresh[0] != {} and fresh[0]._pending is None
    more = model.postprocess_detections(split, [anchors] * cfg.batch, shapes)
    assert more[0].setdefault("scores", None) is not None and more[0].setdefault("extra", 5) == 5
    merged = more[1] | {"image_id": 3}
    assert sorted(merged) == ["boxes", "image_id", "labels", "scores"]
    assert sorted({"image_id": 3} | model.postprocess_detections(split, [anchors] * cfg.batch, shapes)[1]) == sorted(merged)
    third = model.postprocess_detections(split, [anchors] * cfg.batch, shapes)
    del third[0]["labels"]
    assert sorted(third[0]) == ["boxes", "scores"] and list(reversed(third[1])) == ["labels", "scores", "boxes"]
    third[1].clear()
    assert len(third[1]) == 0 and "boxes" not in third[1]
    last = model.postprocess_detections(split, [anchors] * cfg.batch, shapes)
    assert last[0].popitem()[0] == "labels"
    last[1] |= {"k": 1}
    assert len(last[1]) == 4


def test_fused_classification_loss_forward_and_backward(device):
    """Fused focal-loss kernels against the torch restatement of RetinaNetClassificationHead.compute_loss (fp32 on the
    CPU): loss within 1e-5 relative, gradient within 1e-5 of its largest entry plus 1e-4 relative per element."""
    cfg = dataclasses.replace(synthetic.CFG1, num_gt=40, cfg_id=77)
    model = _model(cfg, device)
    anchors_l = _anchors(model, cfg, device)
    anchors = torch.cat(anchors_l).cpu()
    head = synthetic.make_head_outputs(cfg)
    targets = synthetic.make_targets(cfg, empty_image=1)
    matched = tv.retinanet_match([t["boxes"] for t in targets], [anchors] * cfg.batch, cfg.fg_iou_thresh, cfg.bg_iou_thresh)
    # reference on the CPU
    logits_ref = head["cls_logits"].clone().requires_grad_(True)
    ref = tv.retinanet_head_loss(targets, {"cls_logits": logits_ref, "bbox_regression": head["bbox_regression"]},
                                 [anchors] * cfg.batch, matched)["classification"]
    (ref * 3.0).backward()
    # fused kernels through the head object
    logits = head["cls_logits"].to(device).requires_grad_(True)
    t_dev = [{k: v.to(device) for k, v in t.items()} for t in targets]
    m_dev = list(torch.stack(matched).to(device).unbind(0))
    loss = model.head.classification_head.compute_loss(t_dev, {"cls_logits": logits}, m_dev)
    assert loss.grad_fn is not None and type(loss.grad_fn).__name__.startswith("_RetinaNetFocalLoss")
    (loss * 3.0).backward()
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item()), (loss.item(), ref.item())
    g, gr = logits.grad.cpu(), logits_ref.grad
    tol = 1e-5 * gr.abs().max() + 1e-4 * gr.abs()
    assert bool(((g - gr).abs() <= tol).all()), float((g - gr).abs().max())
    # ignored anchors (-2) and the image without targets get exactly what the reference gives them
    assert torch.equal(g[0][matched[0] == -2], torch.zeros_like(g[0][matched[0] == -2]))
    assert int((matched[0] == -2).sum()) > 0
    # another gamma / alpha pair goes through the general power path
    from object_detection_amd import det_utils
    loss2 = det_utils._RetinaNetFocalLoss.apply(logits.detach(), torch.stack(matched).to(device),
                                                torch.cat([t["labels"] for t in t_dev]),
                                                kernels._device_ints([0, 40, 40], device), 0.4, 1.5)
    ref2 = sum(tv.sigmoid_focal_loss(head["cls_logits"][i][matched[i] != -2],
                                     _targets_of(head["cls_logits"][i], matched[i], targets[i]["labels"])[matched[i] != -2],
                                     alpha=0.4, gamma=1.5, reduction="sum") / max(1, int((matched[i] >= 0).sum()))
               for i in range(cfg.batch)) / cfg.batch
    assert abs(loss2.item() - float(ref2)) <= 1e-5 * abs(float(ref2)), (loss2.item(), float(ref2))


def test_fused_regression_loss_through_the_head(device):
    """RetinaNetRegressionHead.compute_loss on the GPU goes through the fused kernels ("giou", the type the reference
    selects) and agrees with the torch restatement on the CPU: loss 1e-5 relative, gradient 1e-5 of its largest entry
    plus 1e-4 relative per element."""
    cfg = dataclasses.replace(synthetic.CFG1, num_gt=40, cfg_id=78)
    model = _model(cfg, device)
    assert model.head.regression_head._loss_type == "giou"
    anchors_l = _anchors(model, cfg, device)
    anchors_dev = torch.cat(anchors_l)
    anchors = anchors_dev.cpu()
    head = synthetic.make_head_outputs(cfg)
    targets = synthetic.make_targets(cfg, empty_image=1)
    matched = tv.retinanet_match([t["boxes"] for t in targets], [anchors] * cfg.batch, cfg.fg_iou_thresh, cfg.bg_iou_thresh)
    reg_ref = head["bbox_regression"].clone().requires_grad_(True)
    ref = tv.retinanet_head_loss_regression(targets, reg_ref, [anchors] * cfg.batch, matched)
    (ref * 0.5).backward()
    reg = head["bbox_regression"].to(device).requires_grad_(True)
    t_dev = [{k: v.to(device) for k, v in t.items()} for t in targets]
    m_dev = list(torch.stack(matched).to(device).unbind(0))
    loss = model.head.regression_head.compute_loss(t_dev, {"bbox_regression": reg}, [anchors_dev] * cfg.batch, m_dev)
    assert loss.grad_fn is not None and type(loss.grad_fn).__name__.startswith("_RetinaNetBoxLoss")
    (loss * 0.5).backward()
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item()), (loss.item(), ref.item())
    g, gr = reg.grad.cpu(), reg_ref.grad
    tol = 1e-5 * gr.abs().max() + 1e-4 * gr.abs()
    assert bool(((g - gr).abs() <= tol).all()), float((g - gr).abs().max())
    assert int((torch.stack(matched) >= 0).sum()) > 0
    # the whole training loss dictionary of the model
    losses = model.compute_loss(t_dev, {"cls_logits": head["cls_logits"].to(device), "bbox_regression": reg.detach()},
                                [anchors_dev] * cfg.batch)
    both = tv.retinanet_head_loss(targets, head, [anchors] * cfg.batch, matched)
    for k in ("classification", "bbox_regression"):
        assert abs(losses[k].item() - both[k].item()) <= 1e-5 * abs(both[k].item()), k


def _targets_of(logits, matched, labels):
    t = torch.zeros_like(logits)
    fg = matched >= 0
    t[fg, labels[matched[fg]]] = 1.0
    return t


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_fused_losses_take_autocast_head_outputs(device, dtype):
    """Head outputs in the 16-bit types autocast training produces go through the fused kernels as they are: the loss
    equals the fp32 evaluation of the torch restatement on the same (already rounded) values within 1e-5, and the
    gradients, returned in the 16-bit type, equal its gradients within one rounding of that type."""
    cfg = dataclasses.replace(synthetic.CFG1, num_gt=40, cfg_id=79)
    model = _model(cfg, device)
    anchors_dev = torch.cat(_anchors(model, cfg, device))
    anchors_half = anchors_dev.to(dtype)                      # AnchorGenerator casts to the feature dtype
    anchors = anchors_half.float().cpu()
    head = synthetic.make_head_outputs(cfg)
    cls16, reg16 = head["cls_logits"].to(dtype), head["bbox_regression"].to(dtype)
    targets = synthetic.make_targets(cfg, empty_image=1)
    matched = tv.retinanet_match([t["boxes"] for t in targets], [anchors] * cfg.batch, cfg.fg_iou_thresh, cfg.bg_iou_thresh)
    cls_ref = cls16.float().requires_grad_(True)
    reg_ref = reg16.float().requires_grad_(True)
    eps = float(torch.finfo(dtype).eps)
    ref_c = tv.retinanet_head_loss(targets, {"cls_logits": cls_ref, "bbox_regression": reg_ref.detach()},
                                   [anchors] * cfg.batch, matched)["classification"]
    ref_r = tv.retinanet_head_loss_regression(targets, reg_ref, [anchors] * cfg.batch, matched, eps=eps)
    (ref_c + ref_r).backward()
    cls_d, reg_d = cls16.to(device).requires_grad_(True), reg16.to(device).requires_grad_(True)
    t_dev = [{k: v.to(device) for k, v in t.items()} for t in targets]
    m_dev = list(torch.stack(matched).to(device).unbind(0))
    losses = model.head.compute_loss(t_dev, {"cls_logits": cls_d, "bbox_regression": reg_d}, [anchors_half] * cfg.batch, m_dev)
    for k in losses:
        assert losses[k].dtype is torch.float32 and type(losses[k].grad_fn).__name__.startswith("_RetinaNet"), k
    assert abs(losses["classification"].item() - ref_c.item()) <= 1e-5 * abs(ref_c.item())
    assert abs(losses["bbox_regression"].item() - ref_r.item()) <= 1e-5 * abs(ref_r.item())
    (losses["classification"] + losses["bbox_regression"]).backward()
    ulp = float(torch.finfo(dtype).eps)
    for got, want in ((cls_d.grad, cls_ref.grad), (reg_d.grad, reg_ref.grad)):
        assert got.dtype is dtype
        g, w = got.float().cpu(), want
        spacing = 2.0 ** -24 if dtype is torch.float16 else 0.0      # fp16 subnormal spacing (no loss scaling here)
        tol = 1e-5 * w.abs().max() + ulp * w.abs() + spacing
        assert bool(((g - w).abs() <= tol).all()), float((g - w).abs().max())


def test_prediction_loop_tail_on_model_detections(device):
    """predict.rescale_nms on the list postprocess_detections returns (padded tensors used where they are, nobody
    waits for the counts) and on materialised dictionaries, against the restatement of reference predict.py:66-76 run
    image by image on the same detections."""
    from object_detection_amd import predict
    from object_detection_amd.retinanet import Detections
    cfg = synthetic.CFG1
    model = _model(cfg, device)
    anchors = _anchors(model, cfg, device)
    sizes = synthetic.level_sizes(cfg)
    head = synthetic.make_head_outputs(cfg)
    shapes = [(512, 512)] * cfg.batch
    split = {k: [t.to(device) for t in v.split(sizes, dim=1)] for k, v in head.items()}
    model_wh = (512, 512)
    image_wh = [(1920, 1080), (333, 517)]
    plain = [{k: v.cpu().clone() for k, v in d.items()} for d in model.postprocess_detections(split, [anchors] * cfg.batch, shapes)]
    assert all(p["boxes"].shape[0] > 1 for p in plain)
    expect = [tv.predict_tail({k: v.clone() for k, v in p.items()}, wh, model_wh, 0.3, float_rule=True)
              for p, wh in zip(plain, image_wh)]
    assert any(e[0].shape[0] < p["boxes"].shape[0] for e, p in zip(expect, plain))   # the second NMS removes something
    # (1) the lazy list of one postprocess call
    dets = model.postprocess_detections(split, [anchors] * cfg.batch, shapes)
    out = predict.rescale_nms(dets, image_wh, model_wh, 0.3)
    assert all(isinstance(d, Detections) and d._pending is not None for d in dets)   # still nobody looked
    for o, d, p, wh, e in zip(out, dets, plain, image_wh, expect):
        assert o["boxes"].dtype is torch.int32
        assert torch.equal(o["boxes"].cpu(), e[0]) and torch.equal(o["scores"].cpu(), e[1])
        assert torch.equal(o["labels"].cpu(), e[2])
        ratio = torch.tensor([wh[0] / model_wh[0], wh[1] / model_wh[1]] * 2)
        assert torch.equal(d["boxes"].cpu(), p["boxes"] * ratio)                    # mul_ semantics: rescaled in place
    # (2) dictionaries somebody already looked at, lists of unequal length
    mat = [{k: v.to(device) for k, v in p.items()} for p in plain]
    out2 = predict.rescale_nms(mat, image_wh, model_wh, 0.3)
    for o, m, p, wh, e in zip(out2, mat, plain, image_wh, expect):
        assert torch.equal(o["boxes"].cpu(), e[0]) and torch.equal(o["scores"].cpu(), e[1])
        assert torch.equal(o["labels"].cpu(), e[2])
        ratio = torch.tensor([wh[0] / model_wh[0], wh[1] / model_wh[1]] * 2)
        assert torch.equal(m["boxes"].cpu(), p["boxes"] * ratio)


@pytest.mark.parametrize("section,cases", [("match", 60), ("select", 60), ("nms", 150), ("nms_segments", 150),
                                           ("fcos", 100), ("tail", 300), ("focal", 25), ("boxloss", 150)])
def test_random_shapes_against_the_scalar_checker(device, section, cases):
    """A fixed run of the randomised parity tool (tests/fuzz_parity.py): ragged pyramids and GT lists, level sizes from
    1 to 300 000 logits with 1 ... 91 classes, NMS in both branches under both threshold rules, the prediction tail --
    every case bit for bit against oracle/od_oracle.c; the fused head losses against the torch restatement within the
    tolerances the tool states."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(GOLD), "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    bad = fuzz.run_seeds(section, range(1000, 1000 + cases))
    assert not bad, bad[:3]


def test_predict_tail_golden_fixture(device):
    """od_predict_tail on the detections of the postprocess fixture against tests/golden/predict_tail_small.npz."""
    g = np.load(os.path.join(GOLD, "predict_tail_small.npz"))
    p = np.load(os.path.join(GOLD, "postprocess_small.npz"))
    D = max(p["boxes0"].shape[0], p["boxes1"].shape[0])
    pad = lambda a, shape, dt: torch.from_numpy(np.concatenate([a, np.zeros((D - a.shape[0],) + shape, dtype=a.dtype)])).to(dt)
    boxes = torch.stack([pad(p[f"boxes{i}"], (4,), torch.float32) for i in range(2)]).to(device)
    scores = torch.stack([pad(p[f"scores{i}"], (), torch.float32) for i in range(2)]).to(device)
    labels = torch.stack([pad(p[f"labels{i}"], (), torch.int64) for i in range(2)]).to(device)
    counts = torch.tensor([p["boxes0"].shape[0], p["boxes1"].shape[0]], dtype=torch.int32, device=device)
    mw, mh = (int(v) for v in g["model_wh"])
    ratio = torch.tensor([[w / mw, h / mh] * 2 for w, h in g["image_wh"].tolist()], dtype=torch.float32, device=device)
    for rule, mode in (("cuda", _C.NMS_THRESHOLD_FLOAT), ("cpu", 0)):
        b = boxes.clone()
        keep, ob, os_, ol, oc = kernels.predict_tail(b, scores, labels, counts, ratio, 0.3, mode)
        for i in range(2):
            k = int(oc[i])
            assert k == g[f"keep{i}_{rule}"].shape[0]
            assert np.array_equal(keep[i, :k].cpu().numpy(), g[f"keep{i}_{rule}"])
            assert np.array_equal(ob[i, :k].cpu().numpy(), g[f"boxes{i}_{rule}"])
            assert np.array_equal(os_[i, :k].cpu().numpy(), g[f"scores{i}_{rule}"])
            assert np.array_equal(ol[i, :k].cpu().numpy(), g[f"labels{i}_{rule}"])
            assert np.array_equal(b[i, :int(counts[i])].cpu().numpy(), g[f"scaled{i}"])


def test_reference_checkpoint_through_the_model_on_gpu(device, tmp_path):
    """The reference's prediction set-up (object_detection/predict.py:44-61) on the GPU: build the model with its
    keywords, ``torch.load`` the ``{'weights', 'height', 'width'}`` file ``train.py:118-122`` wrote (here with the loader
    that executes nothing), ``model.load_state_dict(params['weights'])`` -- strict, as the reference calls it --
    ``model.to(device)``, ``model.eval()``, ``model(images)``, then the loop's tail.  The detections equal, bit for bit,
    those of the model the file was written from; the head entries carry torchvision's names."""
    from object_detection_amd import checkpoint
    from object_detection_amd.predict import rescale_nms
    kw = dict(backbone="pyramid_stub", layers=[0, 1], out_channels=32, min_size=160, max_size=160, nms_thresh=0.3,
              score_thresh=0.005, detections_per_img=80)
    torch.manual_seed(21)
    trained = get_model(4, **kw).to(device).eval()
    images = [torch.rand(3, 160, 160, device=device), torch.rand(3, 120, 150, device=device),
              torch.rand(3, 97, 131, device=device)]
    pil_sizes = [(int(i.shape[-1]), int(i.shape[-2])) for i in images]           # PIL's img.size is (w, h)
    with torch.no_grad():
        want = [{k: v.clone() for k, v in d.items()} for d in trained(images)]
        want_tail = [{k: v.clone() for k, v in d.items()}
                     for d in rescale_nms([dict(d) for d in trained(images)], pil_sizes, (160, 160))]
    assert sum(d["boxes"].shape[0] for d in want) > 0
    path = str(tmp_path / "model_final.pth")
    checkpoint.save_checkpoint(trained, path, 160, 160)

    params = checkpoint.load_checkpoint(path)                                      # predict.py:46
    assert set(params) == {"weights", "height", "width"}
    names = set(params["weights"])
    for name in ("head.classification_head.conv.0.0.weight", "head.classification_head.cls_logits.weight",
                 "head.classification_head.cls_logits.bias", "head.regression_head.conv.3.0.weight",
                 "head.regression_head.bbox_reg.weight", "head.regression_head.bbox_reg.bias"):
        assert name in names, name
    torch.manual_seed(99)                                                          # other initial weights
    model = get_model(4, **kw)                                                     # predict.py:44-45
    report = model.load_state_dict(params["weights"])                              # predict.py:49 (strict)
    assert not report.missing_keys and not report.unexpected_keys
    model.to(device)                                                               # predict.py:54
    model.eval()
    width, height = params["width"], params["height"]                              # predict.py:51-52
    with torch.no_grad():
        got = model(images)                                                        # predict.py:61
        for d, w in zip(got, want):
            for k in ("boxes", "scores", "labels"):
                assert torch.equal(d[k], w[k]), k
        tail = rescale_nms(model(images), pil_sizes, (width, height))              # predict.py:63-76
    for d, w in zip(tail, want_tail):
        assert d["boxes"].dtype is torch.int32
        for k in ("boxes", "scores", "labels"):
            assert torch.equal(d[k], w[k]), k
    # a checkpoint of another class count is refused by the strict load, as torch does for the reference
    other = get_model(7, **kw)
    with pytest.raises(RuntimeError, match="size mismatch"):
        other.load_state_dict(params["weights"])
