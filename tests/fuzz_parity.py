"""Randomised parity run: the HIP entry points against the scalar C checker (oracle/od_oracle.c) on many random shapes
and value distributions (and, within stated floating-point tolerances, the fused head losses against the torch
restatement) -- matching (ragged pyramids, GT counts 0 ... 600, tiny and huge boxes, duplicates), candidate
selection (level sizes 1 ... 300 000, 1 ... 91 classes, thresholds, K), batched NMS (both branches, both threshold
rules, ties, duplicates) and the prediction tail.  Every failure prints the seed that reproduces it.
Usage: [FUZZ_SECTIONS=match,nms] python tests/fuzz_parity.py [seconds per section] [first seed]"""
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd import _C, kernels  # noqa: E402
from oracle import c_oracle  # noqa: E402

dev = torch.device("cuda:0")
_main = __name__ == "__main__"
BUDGET = float(sys.argv[1]) if _main and len(sys.argv) > 1 else 30.0
SEED0 = int(sys.argv[2]) if _main and len(sys.argv) > 2 else 0
failures = []


def boxes_random(n, gen, extent, wh, integer):
    xy = torch.rand((n, 2), generator=gen) * extent
    sz = torch.rand((n, 2), generator=gen) * wh + 1
    b = torch.cat([xy, xy + sz], dim=1)
    return b.round() if integer else b


def fuzz_match(seed):
    gen = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
    levels = ri(1, 4)
    per_loc = [1, 3, 9][ri(0, 2)]
    parts, hint, first = [], [], 0
    extent = float(ri(64, 1200))
    for l in range(levels):
        gw, gh = ri(1, 160 >> l), ri(1, 160 >> l)
        stride = extent / max(gw, gh)
        ys, xs = torch.meshgrid(torch.arange(gh) * stride, torch.arange(gw) * stride, indexing="ij")
        size = stride * (2 + l)
        base = torch.stack([torch.tensor([-size * (0.5 + 0.3 * a), -size / (1 + a), size * (0.5 + 0.3 * a), size / (1 + a)])
                            for a in range(per_loc)]).round()
        shifts = torch.stack([xs, ys, xs, ys], dim=-1).reshape(-1, 1, 4)
        parts.append((shifts + base[None]).reshape(-1, 4))
        hint.append((first, gw * per_loc, gh))
        first += gw * gh * per_loc
    anchors = torch.cat(parts).float()
    if ri(0, 3) == 0:                                  # ragged tail that no grid describes
        anchors = torch.cat([anchors, boxes_random(ri(1, 300), gen, extent, 60.0, False)])
    B = ri(1, 4)
    gts = []
    for b in range(B):
        m = [0, 1, ri(2, 30), ri(31, 300), ri(301, 600)][ri(0, 4)]
        g = boxes_random(m, gen, extent, [8.0, 80.0, extent][ri(0, 2)], ri(0, 1) == 1)
        if m > 3 and ri(0, 1):
            g[m // 2] = g[0]                           # duplicate GT
        if m > 1 and ri(0, 3) == 0:
            g[-1] = torch.tensor([5 * extent, 5 * extent, 5 * extent + 3, 5 * extent + 4])   # overlaps nothing
        gts.append(g)
    hi = [0.5, 0.6, 0.7][ri(0, 2)]
    lo = [0.3, 0.4, 0.5][ri(0, 2)]
    lq = ri(0, 3) != 0
    use_hint = ri(0, 1) == 1
    groups = (0, 2, 3, 8, 13)[ri(0, 4)]                  # workgroups per image of the restore kernel: never changes a result
    _C.match_restore_groups(groups)
    try:
        out = kernels.match_retinanet(anchors.to(dev), [g.to(dev) for g in gts], hi, lo, lq,
                                      grids=hint if use_hint else ()).cpu().numpy()
    finally:
        _C.match_restore_groups(0)
    for b, g in enumerate(gts):
        ref = c_oracle.match_retinanet(g.numpy(), anchors.numpy(), hi, lo, lq) if g.shape[0] else np.full(anchors.shape[0], -1)
        if not np.array_equal(out[b], ref):
            return f"image {b}: {int((out[b] != ref).sum())} of {ref.size} differ (M={g.shape[0]}, N={anchors.shape[0]}, hint={use_hint}, lq={lq}, groups={groups})"
    return None


def fuzz_select(seed):
    gen = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
    C = [1, 2, 4, 7, 91][ri(0, 4)]
    N = [ri(1, 50), ri(51, 3000), ri(3001, 70000), ri(70001, 300000 // C + 70001)][ri(0, 3)]
    B = ri(1, 3)
    K = [1, ri(2, 100), 1000, ri(1001, 3000)][ri(0, 3)]
    thr = [0.0, 0.05, 0.3, 0.5, 0.9][ri(0, 4)]
    mean = [-4.0, -2.0, 0.0, 2.0][ri(0, 3)]
    logits = torch.randn((B, N, C), generator=gen) * 1.5 + mean
    if ri(0, 1):
        logits[torch.rand((B, N, C), generator=gen) < 0.01] = 20.0        # saturated ties
    if ri(0, 3) == 0:
        logits = logits.round()                                           # many equal scores
    fcos = ri(0, 3) == 0
    ctr = torch.randn((B, N, 1), generator=gen) if fcos else None
    reg = torch.randn((B, N, 4), generator=gen) * 0.5
    anchors = boxes_random(N, gen, 500.0, 60.0, True)
    hw = torch.tensor([[ri(100, 600), ri(100, 600)] for _ in range(B)], dtype=torch.int32, device=dev)
    cb, cs, cl, ci, cc = kernels.select_candidates([logits.to(dev)], [reg.to(dev)], [anchors.to(dev)], hw, thr, K,
                                                   _C.CODER_LINEAR if fcos else _C.CODER_BOXCODER, math.log(1000.0 / 16),
                                                   want_index=True, ctr_levels=[ctr.to(dev)] if fcos else None)
    cs, ci, cc, cb = cs.cpu(), ci.cpu(), cc.cpu(), cb.cpu()
    for b in range(B):
        s_ref, i_ref = c_oracle.filter_topk(logits[b].numpy(), ctr[b].numpy().reshape(-1) if fcos else None, C, thr, K)
        n = int(cc[b, 0])
        if n != len(s_ref):
            return f"image {b}: {n} candidates, checker {len(s_ref)} (N={N}, C={C}, K={K}, thr={thr}, fcos={fcos})"
        if not np.array_equal(ci[b, :n].numpy(), i_ref):
            return f"image {b}: indices differ (N={N}, C={C}, K={K}, thr={thr}, fcos={fcos})"
        if not np.array_equal(cs[b, :n].numpy().view(np.uint32), s_ref.view(np.uint32)):
            return f"image {b}: score bits differ (N={N}, C={C}, K={K}, thr={thr}, fcos={fcos})"
        a_idx = i_ref // C
        if fcos:
            dec = c_oracle.decode_linear(reg[b].numpy()[a_idx], anchors.numpy()[a_idx])
        else:
            dec = c_oracle.decode_single(reg[b].numpy()[a_idx], anchors.numpy()[a_idx], math.log(1000.0 / 16))
        dec = c_oracle.clip(dec, int(hw[b, 0]), int(hw[b, 1]))
        if not np.array_equal(cb[b, :n].numpy().view(np.uint32), np.asarray(dec, dtype=np.float32).view(np.uint32)):
            return f"image {b}: decoded boxes differ (N={N}, C={C}, K={K}, fcos={fcos})"
    return None


def fuzz_nms(seed):
    gen = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
    n = [ri(1, 64), ri(65, 600), ri(601, 3000), ri(3001, 9000)][ri(0, 3)]
    classes = [1, 2, 4, 20, 91][ri(0, 4)]
    extent = [30.0, 200.0, 1000.0][ri(0, 2)]
    boxes = boxes_random(n, gen, extent, [10.0, 80.0][ri(0, 1)], ri(0, 1) == 1)
    if n > 4 and ri(0, 1):
        boxes[n // 2] = boxes[1]
        boxes[n // 3] = boxes[1]
    scores = torch.rand(n, generator=gen)
    if ri(0, 1):
        scores = (scores * 16).round() / 16                                # heavy ties
    labels = torch.randint(0, classes, (n,), generator=gen)
    thr = [0.0, 0.3, 0.5, 0.75][ri(0, 3)]
    mode, omode = [(_C.NMS_TRICK, 0), (_C.NMS_VANILLA, 1)][ri(0, 1)]
    float_rule = ri(0, 1) == 1
    keep_cap = [n, ri(1, n), 100][ri(0, 2)]
    cnt = torch.tensor([[n]], dtype=torch.int32, device=dev)
    matrix = _C.NMS_PREFER_MATRIX if ri(0, 1) else 0      # the chip-wide bit-matrix form: the same results
    ref = c_oracle.batched_nms(boxes.numpy(), scores.numpy(), labels.numpy(), thr, omode, float_rule)[:keep_cap]
    keep, ob, os_, ol, oc = kernels.batched_nms_padded(boxes[None].to(dev), scores[None].to(dev), labels[None].to(dev), cnt,
                                                       n, False, thr, mode | (_C.NMS_THRESHOLD_FLOAT if float_rule else 0) | matrix,
                                                       keep_cap)
    k = int(oc.item())
    if k != len(ref) or not np.array_equal(keep[0, :k].cpu().numpy(), ref):
        return f"keep list differs: {k} vs {len(ref)} (n={n}, classes={classes}, thr={thr}, mode={omode}, float={float_rule}, cap={keep_cap}, matrix={matrix})"
    if not torch.equal(ob[0, :k].cpu(), boxes[ref]) or not torch.equal(ol[0, :k].cpu(), labels[ref]):
        return "gathered outputs differ"
    return None


def fuzz_tail(seed):
    gen = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
    B, D = ri(1, 5), [ri(1, 64), ri(65, 300), ri(301, 1024)][ri(0, 2)]
    counts = torch.randint(0, D + 1, (B,), generator=gen, dtype=torch.int32)
    extent = [40.0, 300.0, 1000.0][ri(0, 2)]
    boxes = torch.stack([boxes_random(D, gen, extent, 90.0, ri(0, 1) == 1) for _ in range(B)])
    scores = torch.rand((B, D), generator=gen)
    if ri(0, 1):
        scores = (scores * 8).round() / 8
    if ri(0, 1):
        scores = scores.sort(dim=1, descending=True, stable=True)[0]
    labels = torch.randint(0, 5, (B, D), generator=gen)
    ratio = torch.rand((B, 2), generator=gen).mul(3).add(0.2).repeat(1, 2)
    thr = [0.3, 0.5][ri(0, 1)]
    float_rule = ri(0, 1) == 1
    keep, ob, os_, ol, oc = kernels.predict_tail(boxes.to(dev), scores.to(dev), labels.to(dev), counts.to(dev), ratio.to(dev),
                                                 thr, _C.NMS_THRESHOLD_FLOAT if float_rule else 0, scale_in_place=False)
    for b in range(B):
        c = int(counts[b])
        scaled = boxes[b, :c].numpy() * ratio[b].numpy()
        ref = c_oracle.nms(scaled, scores[b, :c].numpy(), thr, float_rule) if c else np.zeros(0, dtype=np.int64)
        k = int(oc[b].item())
        if k != len(ref) or not np.array_equal(keep[b, :k].cpu().numpy(), ref):
            return f"image {b}: keep list differs ({k} vs {len(ref)}; D={D}, count={c}, thr={thr}, float={float_rule})"
        if not np.array_equal(ob[b, :k].cpu().numpy(), np.trunc(scaled[ref]).astype(np.int32)):
            return f"image {b}: int32 boxes differ"
    return None


def fuzz_nms_segments(seed):
    """Pre-sorted padded segments with ragged counts (what the candidate selection hands to the NMS): merged lazily
    for up to 8 segments, ranked first beyond that."""
    gen = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
    B = ri(1, 3)
    num_seg = [1, 2, 5, 6, 8, 12][ri(0, 5)]
    seg_stride = [ri(1, 40), ri(41, 400), ri(401, 1500)][ri(0, 2)]
    n = num_seg * seg_stride
    classes = [1, 3, 4, 91][ri(0, 3)]
    extent = [60.0, 400.0, 1500.0][ri(0, 2)]
    boxes = torch.stack([boxes_random(n, gen, extent, 80.0, ri(0, 1) == 1) for _ in range(B)])
    scores = torch.rand((B, n), generator=gen)
    if ri(0, 1):
        scores = (scores * 32).round() / 32
    scores = scores.view(B, num_seg, seg_stride).sort(dim=2, descending=True, stable=True)[0].reshape(B, n)
    labels = torch.randint(0, classes, (B, n), generator=gen, dtype=torch.int32)
    counts = torch.randint(0, seg_stride + 1, (B, num_seg), generator=gen, dtype=torch.int32)
    if ri(0, 1):
        counts[0, 0] = seg_stride
    thr = [0.0, 0.3, 0.5, 0.8][ri(0, 3)]
    mode, omode = [(_C.NMS_TRICK, 0), (_C.NMS_VANILLA, 1)][ri(0, 1)]
    float_rule = ri(0, 1) == 1
    max_keep = [1, ri(2, 100), 500, n][ri(0, 3)]
    matrix = _C.NMS_PREFER_MATRIX if ri(0, 1) else 0
    keep, ob, os_, ol, oc = kernels.batched_nms_padded(boxes.to(dev), scores.to(dev), labels.to(dev), counts.to(dev), seg_stride,
                                                       True, thr, mode | (_C.NMS_THRESHOLD_FLOAT if float_rule else 0) | matrix, max_keep)
    for b in range(B):
        valid = (torch.arange(seg_stride)[None, :] < counts[b][:, None]).reshape(-1)
        vb, vs, vl = boxes[b][valid], scores[b][valid], labels[b][valid]
        ref = c_oracle.batched_nms(vb.numpy(), vs.numpy(), vl.numpy(), thr, omode, float_rule)[:max_keep] if vb.shape[0] \
            else np.zeros(0, dtype=np.int64)
        k = int(oc[b].item())
        if k != len(ref) or not np.array_equal(keep[b, :k].cpu().numpy(), ref):
            return (f"image {b}: keep list differs ({k} vs {len(ref)}; segments {num_seg} x {seg_stride}, classes {classes}, "
                    f"thr {thr}, mode {omode}, float {float_rule}, cap {max_keep}, matrix {matrix})")
        if not torch.equal(ob[b, :k].cpu(), vb[ref]) or not torch.equal(os_[b, :k].cpu(), vs[ref]) \
                or not torch.equal(ol[b, :k].cpu(), vl[ref].long()):
            return f"image {b}: gathered outputs differ"
    return None


def fuzz_fcos(seed):
    gen = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
    levels = ri(1, 5)
    extent = float(ri(64, 1024))
    parts = []
    for l in range(levels):
        g = max(1, ri(1, 128) >> l)
        stride = extent / g
        size = float(8 << l)
        ys, xs = torch.meshgrid((torch.arange(g) + 0.5) * stride, (torch.arange(g) + 0.5) * stride, indexing="ij")
        c = torch.stack([xs, ys], dim=-1).reshape(-1, 2)
        parts.append(torch.cat([c - size / 2, c + size / 2], dim=1))
    anchors = torch.cat(parts).float()
    B = ri(1, 4)
    gts = []
    for b in range(B):
        m = [0, 1, ri(2, 40), ri(41, 400)][ri(0, 3)]
        g = boxes_random(m, gen, extent, [10.0, 100.0, extent][ri(0, 2)], ri(0, 1) == 1)
        if m > 2 and ri(0, 1):
            g[1] = g[0]                                # equal areas: first index wins
        gts.append(g)
    radius = [0.5, 1.5, 2.5][ri(0, 2)]
    nf, nl = parts[0].shape[0], parts[-1].shape[0]
    out = kernels.match_fcos(anchors.to(dev), [g.to(dev) for g in gts], nf, nl, radius).cpu().numpy()
    for b, g in enumerate(gts):
        ref = c_oracle.match_fcos(g.numpy(), anchors.numpy(), nf, nl, radius) if g.shape[0] else np.full(anchors.shape[0], -1)
        if not np.array_equal(out[b], ref):
            return f"image {b}: {int((out[b] != ref).sum())} of {ref.size} differ (M={g.shape[0]}, levels={levels}, radius={radius})"
    return None


def fuzz_focal(seed):
    """Fused classification loss, forward and backward, against sigmoid_focal_loss evaluated in fp64 (loss 1e-5
    relative, gradient 1e-5 of its largest entry plus 1e-4 relative per element): class counts 1 ... 91, ragged sizes,
    image strides that break 16-byte alignment, fp32 / fp16 / bf16 logits."""
    from oracle import tv_reference as tv
    gen = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
    C = [1, 2, 3, 4, 5, 8, 20, 91][ri(0, 7)]
    n = [ri(1, 40), ri(41, 3000), ri(3001, 20000)][ri(0, 2)]
    B, M = ri(1, 3), ri(1, 12)
    pad = ri(0, 7)
    dtype = [torch.float32, torch.float32, torch.float16, torch.bfloat16][ri(0, 3)]
    storage = (torch.randn((B, n * C + pad), generator=gen) * 2.0 - 2.0).to(dtype)
    logits = storage[:, :n * C].view(B, n, C)
    matched = torch.randint(-2, M, (B, n), generator=gen)
    if B > 1:
        matched[1] = torch.where(matched[1] >= 0, torch.full_like(matched[1], -1), matched[1])
    labels = torch.randint(0, C, (B * M,), generator=gen)
    offs = [b * M for b in range(B + 1)]
    alpha, gamma = [(0.25, 2.0), (-1.0, 2.0), (0.6, 0.5), (0.25, 0.0)][ri(0, 3)]
    x = logits.double().requires_grad_(True)
    total = 0.0
    for b in range(B):
        t = torch.zeros((n, C), dtype=torch.float64)
        fg = matched[b] >= 0
        t[fg, labels[offs[b]:offs[b + 1]][matched[b][fg]]] = 1.0
        valid = matched[b] != -2
        total = total + tv.sigmoid_focal_loss(x[b][valid], t[valid], alpha=alpha, gamma=gamma, reduction="sum") / max(1, int(fg.sum()))
    ref = total / B
    ref.backward()
    xd = storage.to(dev)[:, :n * C].view(B, n, C)
    md, ld, od = matched.to(dev), labels.to(dev), kernels._device_ints(offs, dev)
    loss, _, nfg = kernels.focal_loss_forward(xd, md, ld, od, alpha, gamma)
    what = f"(C={C}, n={n}, B={B}, pad={pad}, {dtype}, alpha={alpha}, gamma={gamma})"
    if nfg.cpu().tolist() != [int((matched[b] >= 0).sum()) for b in range(B)]:
        return "foreground counts differ " + what
    if abs(loss.item() - ref.item()) > 1e-5 * abs(ref.item()) + 1e-7:
        return f"loss {loss.item()} vs {ref.item()} " + what
    grad = kernels.focal_loss_backward(xd, md, ld, od, alpha, gamma, nfg, torch.ones((1,), device=dev)).cpu().float()
    gr = x.grad.float()
    rel = {torch.float32: 1e-4, torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]      # one rounding of the stored gradient
    # fp16 gradients of a long, sparsely matched image reach the subnormal range, where the stored value moves in
    # steps of 2^-24
    tol = 1e-5 * gr.abs().max() + rel * gr.abs() + (2.0 ** -24 if dtype is torch.float16 else 0.0)
    if not bool(((grad - gr).abs() <= tol).all()):
        return f"gradient off by {float((grad - gr).abs().max())} " + what
    if not torch.equal(grad[matched == -2], torch.zeros_like(grad[matched == -2])):
        return "gradient of ignored anchors is not zero " + what
    return None


def fuzz_boxloss(seed):
    """Fused regression loss (giou / l1 / smooth_l1), forward and backward, against the fp32 torch restatement."""
    from oracle import tv_reference as tv
    gen = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
    loss_type = ["giou", "l1", "smooth_l1"][ri(0, 2)]
    n = [ri(1, 50), ri(51, 3000), ri(3001, 12000)][ri(0, 2)]
    B, M = ri(1, 3), ri(1, 12)
    shared = ri(0, 1) == 1
    weights = (1.0, 1.0, 1.0, 1.0) if ri(0, 1) else (10.0, 10.0, 5.0, 5.0)
    ub = lambda c: torch.cat([(xy := torch.rand((c, 2), generator=gen) * 900.0), xy + 8.0 + torch.rand((c, 2), generator=gen) * 192.0], dim=1)
    anchors = ub(n) if shared else torch.stack([ub(n) for _ in range(B)])
    gt = [ub(M) for _ in range(B)]
    matched = torch.randint(-2, M, (B, n), generator=gen)
    if B > 1:
        matched[B - 1] = torch.where(matched[B - 1] >= 0, torch.full_like(matched[B - 1], -1), matched[B - 1])
    reg = torch.randn((B, n, 4), generator=gen) * 0.5
    reg[0, ::97, 2] = 5.0
    coder = tv.BoxCoder(weights)
    eps = float(torch.finfo(torch.float32).eps)
    x = reg.clone().requires_grad_(True)
    a_list = [(anchors if shared else anchors[b]) for b in range(B)]
    ref = tv.retinanet_head_loss_regression([{"boxes": b} for b in gt], x, a_list, list(matched), loss_type, 0.3, coder, eps=eps)
    ref.backward()
    code = {"l1": _C.BOX_LOSS_L1, "smooth_l1": _C.BOX_LOSS_SMOOTH_L1, "giou": _C.BOX_LOSS_GIOU}[loss_type]
    conf = (code, 0.3, weights, coder.bbox_xform_clip, eps)
    rd, ad, md = reg.to(dev), anchors.to(dev), matched.to(dev)
    gd, od = torch.cat(gt).to(dev), kernels._device_ints([b * M for b in range(B + 1)], dev)
    loss, _, nfg = kernels.box_loss_forward(rd, ad, md, gd, od, *conf)
    what = f"({loss_type}, n={n}, B={B}, M={M}, shared={shared}, weights={weights})"
    if abs(loss.item() - ref.item()) > 1e-5 * abs(ref.item()) + 1e-7:
        return f"loss {loss.item()} vs {ref.item()} " + what
    grad = kernels.box_loss_backward(rd, ad, md, gd, od, *conf, nfg, torch.full((1,), 2.0, device=dev)).cpu()
    gr = 2.0 * x.grad
    tol = 1e-5 * gr.abs().max() + 1e-4 * gr.abs()
    wrong = (grad - gr).abs() > tol
    if loss_type == "l1" and bool(wrong.any()):
        # the l1 gradient is the sign of (regression - encoded target): where the two agree to the last bits the sign
        # hangs on the rounding of the encoder's logarithm, and either answer is a subgradient
        for b, a, c in wrong.nonzero().tolist():
            t = coder.encode_single(gt[b][matched[b, a]][None], a_list[b][a][None])[0, c]
            if abs(float(reg[b, a, c] - t)) <= 4e-7 * max(1.0, abs(float(t))):
                wrong[b, a, c] = False
    if bool(wrong.any()):
        return f"gradient off by {float((grad - gr).abs()[wrong].max())} " + what
    if not torch.equal(grad[matched < 0], torch.zeros_like(grad[matched < 0])):
        return "gradient of anchors that are not foreground is not zero " + what
    return None


SECTIONS = (("match", fuzz_match), ("select", fuzz_select), ("nms", fuzz_nms), ("nms_segments", fuzz_nms_segments),
            ("fcos", fuzz_fcos), ("tail", fuzz_tail), ("focal", fuzz_focal), ("boxloss", fuzz_boxloss))


def run_seeds(name, seeds):
    """Messages of the failing seeds of one section (empty list: all cases agree with the checker)."""
    fn = dict(SECTIONS)[name]
    bad = []
    for seed in seeds:
        try:
            msg = fn(seed)
        except Exception as exc:  # noqa: BLE001 -- a launcher error is a finding too
            msg = f"exception {type(exc).__name__}: {exc}"
        if msg:
            bad.append(f"{name} seed {seed}: {msg}")
    return bad


if __name__ == "__main__":
    if os.environ.get("FUZZ_SELFTEST"):
        # the harness must notice a wrong answer: the checker is asked for a different threshold than the kernel
        real = c_oracle.nms
        c_oracle.nms = lambda b, s, t, f=False: real(b, s, t + 0.05, f)
        assert run_seeds("tail", range(20)), "self-test: a perturbed checker went unnoticed"
        c_oracle.nms = real
        print("self-test ok: a perturbed checker is noticed")
    only = [n for n in os.environ.get("FUZZ_SECTIONS", "").split(",") if n]      # e.g. FUZZ_SECTIONS=match,nms
    for name, _ in SECTIONS:
        if only and name not in only:
            continue
        t0, cases, seed, bad = time.time(), 0, SEED0, []
        while time.time() - t0 < BUDGET and len(bad) < 5:
            found = run_seeds(name, [seed])
            for msg in found:
                print("FAIL", msg, flush=True)
            bad += found
            cases += 1
            seed += 1
        failures += bad
        print(f"{name}: {cases} random cases in {time.time() - t0:.0f} s, {len(bad)} failures", flush=True)
    sys.exit(1 if failures else 0)
