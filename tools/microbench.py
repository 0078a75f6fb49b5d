"""Per-kernel timings on the GPU (HIP events inside the library) for quick A/B while tuning."""
import argparse
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


REPS, WARM = 20, 3


def timed(fn, reps=None, warm=None):
    reps = REPS if reps is None else reps
    warm = WARM if warm is None else warm
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    _C.profile_select(_C.PROFILE_ALL)
    for k in range(len(_C.KERNEL_NAMES)):
        _C.profile_read(k)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    out = {}
    for k, name in enumerate(_C.KERNEL_NAMES):
        ms, n = _C.profile_read(k)
        if n:
            out[name] = round(1e3 * ms / n, 2)
    _C.profile_select(_C.PROFILE_OFF)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="all")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--gt", type=int, default=100)
    ap.add_argument("--levels", type=int, default=6, help="use only the finest LEVELS pyramid levels")
    ap.add_argument("--logits", default="prior", choices=sorted(synthetic.LOGIT_DISTRIBUTIONS))
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--lib", default=None, help="experimental build of libod_boxes.so to load instead")
    args = ap.parse_args()
    global REPS, WARM
    REPS = args.reps
    WARM = min(WARM, args.reps)
    if args.lib:
        _C.LIB_PATH = os.path.abspath(args.lib)
    dev = torch.device("cuda:0")
    cfg = dataclasses.replace(synthetic.CFG2, batch=args.batch, num_gt=args.gt, grids=synthetic.CFG2.grids[:args.levels])
    sizes = synthetic.level_sizes(cfg)
    ag = AnchorGenerator(synthetic.retinanet_anchor_sizes(len(cfg.grids)), ((0.5, 1.0, 2.0),) * len(cfg.grids))
    grids = [(g, g) for g in cfg.grids]
    strides = [(cfg.image_size[0] // g, cfg.image_size[1] // g) for g in cfg.grids]
    anchors = kernels.anchor_generate(grids, strides, ag.cell_anchors, dev)
    res = {}
    if args.what in ("all", "match"):
        targets = synthetic.make_targets(cfg, empty_image=cfg.batch - 1)
        gts = [t["boxes"].to(dev) for t in targets]
        res["match"] = timed(lambda: kernels.match_retinanet(anchors, gts, 0.6, 0.5, True))
        # debug: device counters of the last call (layout mirrors match_ws_layout in od_match.hip)
        a256 = lambda v: (v + 255) & ~255
        m_total = sum(int(g.shape[0]) for g in gts)
        n = anchors.shape[-2]
        tiles = len(gts) * (((n + 255) // 256) * 5 // 4 + 8 * 9)      # the capacity the workspace is laid out for
        off = a256(m_total * 4 * 32) + a256(len(gts) * n * 4) + a256(tiles * 4) + a256(tiles * 48 * 8) + a256(len(gts) * 4)
        ws = kernels._workspaces[(anchors.device.index, torch.cuda.current_stream(anchors.device).cuda_stream, "match")]
        hc = ws[off + 256:off + 256 + len(gts) * 128].view(torch.int64)[::16]
        res["match_counters"] = {"heavy_tiles": int((hc >> 32).sum()), "heavy_entries": int((hc & 0xffffffff).sum()),
                                 "tiles": tiles}
    if args.what in ("all", "post", "nms", "select", "select_only"):
        head = synthetic.make_head_outputs(cfg, logits=args.logits)
        cls = head["cls_logits"].to(dev)
        reg = head["bbox_regression"].to(dev)
        cls_l, reg_l, anc_l = list(cls.split(sizes, 1)), list(reg.split(sizes, 1)), list(anchors.split(sizes))
        hw = _image_hw_tensor([cfg.image_size] * cfg.batch, dev)
        K = cfg.topk_candidates
        sel = lambda: kernels.select_candidates(cls_l, reg_l, anc_l, hw, cfg.score_thresh, K, _C.CODER_BOXCODER, 4.135)
        res["select"] = timed(sel)
        if args.what == "select_only":        # builds whose select kernel stops early leave the candidates unwritten
            print(json.dumps(res))
            return
        A = 9

        def to_maps(levels, per_anchor):      # [B, g*g*A, k] -> the convolution layout [B, A*k, g, g]
            return [t.view(cfg.batch, g, g, A, per_anchor).permute(0, 3, 4, 1, 2).reshape(cfg.batch, A * per_anchor, g, g)
                    .contiguous() for t, g in zip(levels, cfg.grids)]

        cls_m, reg_m = to_maps(cls_l, cls.shape[-1]), to_maps(reg_l, 4)
        res["select_planar"] = timed(lambda: kernels.select_candidates_planar(
            cls_m, reg_m, anc_l, hw, cls.shape[-1], cfg.score_thresh, K, _C.CODER_BOXCODER, 4.135))
        pl = kernels.select_candidates_planar(cls_m, reg_m, anc_l, hw, cls.shape[-1], cfg.score_thresh, K,
                                              _C.CODER_BOXCODER, 4.135)
        ref = sel()
        valid = (torch.arange(K, device=dev)[None, None, :] < ref[4][:, :, None]).reshape(cfg.batch, -1)   # slots past a count are not written
        res["planar_equal"] = bool(torch.equal(pl[4], ref[4])) and all(bool(torch.equal(a[valid], b[valid]))
                                                                      for a, b in zip(pl[:3], ref[:3]))
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            [m.view(cfg.batch, A, -1, g, g).permute(0, 3, 4, 1, 2).reshape(cfg.batch, -1, m.shape[1] // A)
             for m, g in zip(cls_m + reg_m, cfg.grids + cfg.grids)]
        t1.record(); torch.cuda.synchronize()
        res["permute_copies_us"] = round(t0.elapsed_time(t1) * 100.0, 1)
        cb, cs, cl, _, cc = sel()
        res["cand_per_image"] = float(cc.sum().item()) / cfg.batch
        for mk in (1, 100, 500, 6000):
            res[f"nms_keep{mk}"] = timed(lambda: kernels.batched_nms_padded(cb, cs, cl, cc, K, True, cfg.nms_thresh,
                                                                           _C.NMS_AUTO_GPU, mk, want_keep_idx=False))
        for thr in (0.05, 0.9):
            res[f"nms_thr{thr}_keep500"] = timed(lambda: kernels.batched_nms_padded(cb, cs, cl, cc, K, True, thr,
                                                                                   _C.NMS_AUTO_GPU, 500, want_keep_idx=False))
    if args.what in ("all", "loss"):
        head = synthetic.make_head_outputs(cfg, logits=args.logits)
        cls = head["cls_logits"].to(dev)
        targets = synthetic.make_targets(cfg, empty_image=cfg.batch - 1)
        gts = [t["boxes"].to(dev) for t in targets]
        labels = [t["labels"].to(dev) for t in targets]
        matched = kernels.match_retinanet(anchors, gts, 0.6, 0.5, True)
        offs = [0]
        for lab in labels:
            offs.append(offs[-1] + lab.shape[0])
        gl, go = torch.cat(labels), kernels._device_ints(offs, dev)
        one = torch.ones((1,), device=dev)

        def fwd_bwd():
            _, _, nfg = kernels.focal_loss_forward(cls, matched, gl, go, 0.25, 2.0)
            kernels.focal_loss_backward(cls, matched, gl, go, 0.25, 2.0, nfg, one)

        reg = head["bbox_regression"].to(dev)
        gb = torch.cat(gts)
        bconf = (_C.BOX_LOSS_GIOU, 1.0, (1.0, 1.0, 1.0, 1.0), 4.135166556742356, 1.1920928955078125e-07)

        def box_fwd_bwd():
            _, _, nfg = kernels.box_loss_forward(reg, anchors, matched, gb, go, *bconf)
            kernels.box_loss_backward(reg, anchors, matched, gb, go, *bconf, nfg, one)

        res["box_loss"] = timed(box_fwd_bwd)
        res["foreground_per_image"] = float((matched >= 0).sum().item()) / cfg.batch
        res["loss"] = timed(fwd_bwd)
        res["loss_forward_only_loop"] = timed(lambda: kernels.focal_loss_forward(cls, matched, gl, go, 0.25, 2.0))
        E = cls.numel()
        res["loss_GBps"] = {"focal_fwd": (4 * E + 8 * matched.numel()) / (res["loss"]["focal_fwd"] * 1e-6) / 1e9,
                            "focal_bwd": (8 * E + 8 * matched.numel()) / (res["loss"]["focal_bwd"] * 1e-6) / 1e9}
    if args.what in ("all", "train"):
        # training side of the model: compute_loss forward + backward for the batch, fused kernels against the same
        # steps as torch ops on the GPU (per-image loop, as torchvision writes it)
        import time
        from object_detection_amd.retinanet_model import get_model
        L = len(cfg.grids)
        model = get_model(cfg.num_classes, layers=list(range(L - 2)), backbone="pyramid_stub", out_channels=32,
                          fg_iou_thresh=cfg.fg_iou_thresh, bg_iou_thresh=cfg.bg_iou_thresh).to(dev)
        head = {k: v.to(dev).requires_grad_(True) for k, v in synthetic.make_head_outputs(cfg, logits=args.logits).items()}
        targets = [{k: v.to(dev) for k, v in t.items()} for t in synthetic.make_targets(cfg, empty_image=cfg.batch - 1)]
        apl = [anchors] * cfg.batch

        def fused_step():
            for v in head.values():               # leaves here (in a model the head outputs are not): without this the
                v.grad = None                     # step would also time the accumulation into 400 MB of old gradients
            losses = model.compute_loss(targets, head, apl)
            (losses["classification"] + losses["bbox_regression"]).backward()

        def focal(x, t):                          # sigmoid_focal_loss, alpha 0.25, gamma 2, summed
            p = torch.sigmoid(x)
            ce = torch.nn.functional.binary_cross_entropy_with_logits(x, t, reduction="none")
            p_t = p * t + (1 - p) * (1 - t)
            return ((0.25 * t + 0.75 * (1 - t)) * ce * (1 - p_t) ** 2).sum()

        def giou_sum(reg, anc, gt):               # decode_single + generalized_box_iou_loss, summed
            w, h = anc[:, 2] - anc[:, 0], anc[:, 3] - anc[:, 1]
            cx, cy = anc[:, 0] + 0.5 * w, anc[:, 1] + 0.5 * h
            pcx, pcy = reg[:, 0] * w + cx, reg[:, 1] * h + cy
            pw = torch.exp(reg[:, 2].clamp(max=4.135166556742356)) * w
            ph = torch.exp(reg[:, 3].clamp(max=4.135166556742356)) * h
            x1, y1, x2, y2 = pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph
            xk1, yk1 = torch.max(x1, gt[:, 0]), torch.max(y1, gt[:, 1])
            xk2, yk2 = torch.min(x2, gt[:, 2]), torch.min(y2, gt[:, 3])
            inter = ((xk2 - xk1).clamp(min=0) * (yk2 - yk1).clamp(min=0))
            union = (x2 - x1) * (y2 - y1) + (gt[:, 2] - gt[:, 0]) * (gt[:, 3] - gt[:, 1]) - inter
            area_c = (torch.max(x2, gt[:, 2]) - torch.min(x1, gt[:, 0])) * (torch.max(y2, gt[:, 3]) - torch.min(y1, gt[:, 1]))
            return (1 - inter / (union + 1e-7) + (area_c - union) / (area_c + 1e-7)).sum()

        def torch_step():
            for v in head.values():
                v.grad = None
            matched = model.match_anchors(targets, apl)
            cls_losses, reg_losses = [], []
            for t, logits, reg, m in zip(targets, head["cls_logits"], head["bbox_regression"], matched):
                fg = m >= 0
                tgt = torch.zeros_like(logits)
                tgt[fg, t["labels"][m[fg]]] = 1.0
                valid = m != -2
                cls_losses.append(focal(logits[valid], tgt[valid]) / max(1, int(fg.sum())))
                fi = torch.where(fg)[0]
                reg_losses.append(giou_sum(reg[fi], anchors[fi], t["boxes"][m[fi]]) / max(1, fi.numel()))
            (sum(cls_losses) / cfg.batch + sum(reg_losses) / cfg.batch).backward()

        for name, fn, reps in (("fused", fused_step, 20), ("torch_ops_after_fused_matching", torch_step, 3)):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            res["train_step_us_" + name] = round((time.perf_counter() - t0) / reps * 1e6, 1)
    if args.what in ("all", "pre"):
        from object_detection_amd.transform import GeneralizedRCNNTransform
        tr = GeneralizedRCNNTransform(1024, 1024, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
        for name, shape in (("identity_1024", (1024, 1024)), ("resize_from_720x960", (720, 960))):
            imgs = [torch.rand((3,) + shape, device=dev) for _ in range(cfg.batch)]
            res["pre_" + name] = timed(lambda: tr(imgs))
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            t0.record()
            mean = torch.tensor(tr.image_mean, device=dev)[:, None, None]
            std = torch.tensor(tr.image_std, device=dev)[:, None, None]
            for _ in range(5):                      # the same work as separate torch ops, image by image
                out = []
                for im in imgs:
                    im = (im - mean) / std
                    size = tr._resized_size(*shape)
                    out.append(torch.nn.functional.interpolate(im[None], size=size, mode="bilinear", align_corners=False)[0])
                batch = torch.zeros((len(out), 3, 1024, 1024), device=dev)
                for o, slot in zip(out, batch):
                    slot[:, :o.shape[1], :o.shape[2]].copy_(o)
            t1.record(); torch.cuda.synchronize()
            res["pre_" + name + "_torch_ops_us"] = round(t0.elapsed_time(t1) * 200.0, 1)
    if os.environ.get("OD_STAMPS"):
        kernels.batched_nms_padded(cb, cs, cl, cc, K, True, cfg.nms_thresh, _C.NMS_AUTO_GPU, 500, want_keep_idx=False)
        torch.cuda.synchronize()
        ws = kernels._workspaces[(dev.index, torch.cuda.current_stream(dev).cuda_stream, "nms")]
        off = 3 * ((cfg.batch * cb.shape[1] * 4 + 255) & ~255)       # after the order / kept_out / kept_cls arrays
        st = ws[off:off + cfg.batch * 256].view(torch.int64).view(cfg.batch, 32).cpu()
        # [0] start, [1] extremes of the image, [2..3] heads; then five stamps per strip: order + class positions,
        # A (kept list), compaction, B (pairs), C + append; [30] strips done, [31] outputs written
        for b in (0, 5):
            r = st[b]
            res[f"stamps_us_img{b}"] = [round(float(r[i] - r[0]) / 100.0, 1) for i in [0, 1] + list(range(4, 32))]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
