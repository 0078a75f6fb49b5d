"""Parity of each HIP kernel (through the C ABI) against the CPU checkers, bit-exact for indices."""
import math

import numpy as np
import pytest
import torch

from object_detection_amd import _C, kernels, synthetic
from oracle import c_oracle, tv_reference as tv

pytestmark = pytest.mark.gpu


def _anchors_cpu(image, grids, sizes, ratios=(0.5, 1.0, 2.0)):
    ag = tv.AnchorGenerator(sizes, (ratios,) * len(sizes))
    return ag, ag.forward(image, [(g, g) for g in grids], 1)[0]


def test_anchor_generate(device):
    grids = (64, 32, 16, 8, 4)
    sizes = synthetic.retinanet_anchor_sizes(5, 32)
    ag, ref = _anchors_cpu((512, 512), grids, sizes)
    out = kernels.anchor_generate([(g, g) for g in grids], [(512 // g, 512 // g) for g in grids], ag.cell_anchors, device)
    assert out.shape == ref.shape == (49104, 4)
    assert torch.equal(out.cpu(), ref)
    # non-square grid / stride, single base anchor (FCOS style)
    ag2 = tv.AnchorGenerator(((8,), (16,)), ((1.0,),) * 2)
    ref2 = torch.cat(ag2.grid_anchors([(5, 7), (3, 4)], [(8, 9), (16, 17)]))
    out2 = kernels.anchor_generate([(5, 7), (3, 4)], [(8, 9), (16, 17)], ag2.cell_anchors, device)
    assert torch.equal(out2.cpu(), ref2)


@pytest.mark.parametrize("m", [1, 7, 25, 100, 300, 513, 700])
def test_match_retinanet(device, m):
    grids = (64, 32, 16, 8, 4)
    ag, anchors = _anchors_cpu((512, 512), grids, synthetic.retinanet_anchor_sizes(5, 32))
    gts = []
    for b in range(3):
        gen = torch.Generator().manual_seed(100 * m + b)
        gts.append(synthetic.make_gt_boxes(m if b != 1 else max(0, m - 1), (512, 512), gen))
    gts.append(torch.zeros((0, 4)))  # empty-target image
    out = kernels.match_retinanet(anchors.to(device), [g.to(device) for g in gts], 0.6, 0.5, True).cpu()
    for b, g in enumerate(gts):
        ref = c_oracle.match_retinanet(g.numpy(), anchors.numpy(), 0.6, 0.5, True)
        assert np.array_equal(out[b].numpy(), ref), f"image {b}"
    ref_t = tv.retinanet_match(gts, [anchors] * len(gts), 0.6, 0.5)
    for b in range(len(gts)):
        assert torch.equal(out[b], ref_t[b])
    out_nolq = kernels.match_retinanet(anchors.to(device), [gts[0].to(device)], 0.6, 0.5, False).cpu()
    assert np.array_equal(out_nolq[0].numpy(), c_oracle.match_retinanet(gts[0].numpy(), anchors.numpy(), 0.6, 0.5, False))


def test_match_retinanet_lists_that_overflow(device):
    """The one-launch matching path keeps its intermediate results in fixed-size lists (entries per tile, tiles with
    further entries, entries and ties per workgroup of the restore kernel, overflowed tiles per image); inputs that
    overflow each of them take the slower routes and give the checker's answer all the same."""
    gen = torch.Generator().manual_seed(77)

    def check(anchors, gts, what):
        out = kernels.match_retinanet(anchors.to(device), [g.to(device) for g in gts], 0.6, 0.5, True).cpu()
        for b, g in enumerate(gts):
            ref = c_oracle.match_retinanet(g.numpy(), anchors.numpy(), 0.6, 0.5, True)
            assert np.array_equal(out[b].numpy(), ref), (what, b)

    # (1) every anchor is the same box: every tile ties for the only GT's maximum -- more ties (1200) than a workgroup of
    #     the restore kernel lists, and every anchor is restored
    same = torch.tensor([[10., 10., 50., 40.]]).repeat(300_000, 1)
    check(same, [torch.tensor([[12., 8., 60., 44.]]), torch.tensor([[0., 0., 20., 20.], [300., 300., 310., 310.]])], "ties")
    # (2) the same with 2.1 M anchors: more entries of one GT row (8200) than a workgroup lists
    check(same.repeat(7, 1), [torch.tensor([[12., 8., 60., 44.]])], "entries of a row")
    # (3) ten GT that cover the whole pyramid: 2300 tiles with 10 entries each -- more tiles with further entries than
    #     are listed
    ag, anchors = _anchors_cpu((512, 512), (256, 16), ((16,), (64,)), (1.0,))
    anchors = torch.cat([anchors] * 9)                                        # 592 000 anchors
    wide = torch.tensor([[0., 0., 512., 512.]]).repeat(10, 1) + torch.arange(10).view(10, 1) * torch.tensor([[1., 2., -1., -2.]])
    check(anchors, [wide, wide[:3]], "tiles with further entries")
    # (4) sixty such GT on 100 000 anchors: 390 tiles whose entry slots overflow (47 per light tile) -- more than the
    #     restore kernel lists per image; and 300 GT on a cooperative tile (256 slots)
    ag, small = _anchors_cpu((512, 512), (100, 8), ((16,), (128,)), (1.0,))
    small = torch.cat([small] * 10)
    many = torch.tensor([[0., 0., 512., 512.]]).repeat(60, 1) + torch.rand((60, 1), generator=gen) * torch.tensor([[3., 5., -4., -2.]])
    check(small, [many, torch.cat([many] * 5)], "overflowed tiles")


def test_match_retinanet_restore_groups_knob(device):
    """od_match_restore_groups only changes how the restore work is dealt out: every setting gives the checker's
    answer (power-of-two and other counts, the bounds, 0 = default); values outside [2, 64] are refused."""
    gen = torch.Generator().manual_seed(11)
    _, anchors = _anchors_cpu((512, 512), (128, 64, 32, 16), ((16,), (32,), (64,), (128,)), (0.5, 1.0, 2.0))
    gts = []
    for m in (37, 0, 150):
        xy = (torch.rand((m, 2), generator=gen) * 400).round()
        gts.append(torch.cat([xy, xy + (10 + torch.rand((m, 2), generator=gen) * 120).round()], dim=1))
    want = [c_oracle.match_retinanet(g.numpy(), anchors.numpy(), 0.5, 0.4, True) for g in gts]
    a = anchors.to(device)
    g_dev = [g.to(device) for g in gts]
    assert _C.match_restore_groups(0) == 0
    try:
        for groups in (2, 3, 5, 8, 16, 64, 0):
            _C.match_restore_groups(groups)
            out = kernels.match_retinanet(a, g_dev, 0.5, 0.4, True).cpu()
            for b in range(3):
                assert np.array_equal(out[b].numpy(), want[b]), (groups, b)
        for bad in (1, 65, -3):
            with pytest.raises(_C.OdBoxesError):
                _C.match_restore_groups(bad)
    finally:
        _C.match_restore_groups(0)


def test_packed_target_boxes_are_read_in_place(device):
    """Views handed out by pack_boxes are recognised by the matching and loss launchers (no copy), images without
    boxes at the head, in the middle and at the tail of the batch included; results equal those of separate tensors."""
    gen = torch.Generator().manual_seed(5)
    _, anchors = _anchors_cpu((256, 256), (32, 16), ((32,), (64,)), (0.5, 1.0, 2.0))
    sizes = [0, 3, 0, 5, 1, 0]
    separate = []
    for m in sizes:
        xy = torch.rand((m, 2), generator=gen) * 150
        separate.append(torch.cat([xy, xy + 20 + torch.rand((m, 2), generator=gen) * 80], dim=1).to(device))
    views = kernels.pack_boxes(separate)
    assert [v.shape[0] for v in views] == sizes
    gt, offsets, m_total = kernels._pack_gt(views, torch.device(device))
    assert m_total == sum(sizes) and gt.data_ptr() == views[1].data_ptr() and gt.shape == (m_total, 4)
    assert offsets.cpu().tolist() == [0, 0, 3, 3, 8, 9, 9]
    # a list that is not one buffer in order is copied
    swapped = [views[3], views[1]]
    gt2, _, _ = kernels._pack_gt(swapped, torch.device(device))
    assert gt2.data_ptr() not in (views[1].data_ptr(), views[3].data_ptr())
    assert torch.equal(gt2, torch.cat(swapped))
    a = anchors.to(device)
    assert torch.equal(kernels.match_retinanet(a, views, 0.5, 0.4, True), kernels.match_retinanet(a, separate, 0.5, 0.4, True))


def _grids_of(grids, per_location):
    """(first, row_len, rows) of every level of square pyramid grids with `per_location` anchors per cell."""
    out, first = [], 0
    for g in grids:
        out.append((first, g * per_location, g))
        first += g * g * per_location
    return out


@pytest.mark.parametrize("m", [7, 100, 700])
def test_match_retinanet_anchor_grids(device, m):
    """The hint about the order of the anchor list (od_match_retinanet_grids) changes which anchors share a tile and
    nothing else: same matches as the scalar checker and as the call without the hint -- on power-of-two pyramids with
    9 and with 1 anchor per cell, on pyramids whose row lengths and row counts are not multiples of the tile's 64
    anchors and 4 rows (overlapping tiles), with an anchor count that is not a multiple of the tile size, with a table
    per image, with GT that overlap nothing, with an image without GT."""
    cases = [((512, 512), (64, 32, 16, 8, 4), synthetic.retinanet_anchor_sizes(5, 32), (0.5, 1.0, 2.0)),
             ((512, 512), (128, 64, 5), ((16,), (32,), (64,)), (1.0,)),
             # rows of 900 / 450 / ... anchors and 100 / 50 / ... rows: on the two finest levels the last column and the
             # last band of tiles overlap their neighbours; the small coarse levels are tiled consecutively
             ((800, 800), (100, 50, 25, 13, 7), synthetic.retinanet_anchor_sizes(5, 32), (0.5, 1.0, 2.0)),
             ((1000, 1000), (250, 130), ((16,), (32,)), (1.0,))]
    for image, grids, sizes, ratios in cases:
        ag, anchors = _anchors_cpu(image, grids, sizes, ratios)
        hint = _grids_of(grids, len(sizes[0]) * len(ratios))
        assert hint[-1][0] + hint[-1][1] * hint[-1][2] == anchors.shape[0]
        gts = [synthetic.make_gt_boxes(m if b != 1 else m - 1, image, torch.Generator().manual_seed(31 * m + b)) for b in range(3)]
        gts[2] = torch.cat((gts[2][:3], torch.tensor([[9000., 9000., 9010., 9020.]]), gts[2][3:]))   # restores every anchor
        gts.append(torch.zeros((0, 4)))
        g_d = [g.to(device) for g in gts]
        for lq in (True, False):
            plain = kernels.match_retinanet(anchors.to(device), g_d, 0.6, 0.5, lq, grids=()).cpu()
            # an anchor that no tile covered would keep whatever the output buffer held: make that a value no match can
            # be (the allocator hands the block just freed to the next request of the same size)
            poison = torch.full((len(gts), anchors.shape[0]), -7, dtype=torch.int64, device=device)
            del poison
            hinted = kernels.match_retinanet(anchors.to(device), g_d, 0.6, 0.5, lq, grids=hint)
            assert int(hinted.min()) >= -2
            hinted = hinted.cpu()
            assert torch.equal(plain, hinted), (grids, lq)
            for b, g in enumerate(gts):
                assert np.array_equal(hinted[b].numpy(), c_oracle.match_retinanet(g.numpy(), anchors.numpy(), 0.6, 0.5, lq)), (grids, lq, b)
        tables = torch.stack([anchors * (1.0 + 0.05 * b) + 2.0 * b for b in range(len(gts))])
        out = kernels.match_retinanet(tables.to(device), g_d, 0.6, 0.5, True, grids=hint).cpu()
        for b, g in enumerate(gts):
            assert np.array_equal(out[b].numpy(), c_oracle.match_retinanet(g.numpy(), tables[b].numpy(), 0.6, 0.5, True)), (grids, b)
    # the anchor generator leaves the note on its output, and the matching picks it up
    gen = kernels.anchor_generate([(64, 64), (32, 32)], [(8, 8), (16, 16)], ag.cell_anchors[:2], device)
    assert gen._od_anchor_grids == ((0, 64, 64), (4096, 32, 32))


def test_batched_anchor_tables(device):
    """Anchors given per image ([B, N, 4], anchor_batch_stride = N) instead of one table shared by the batch:
    od_match_retinanet, od_match_fcos and od_select_candidates against the scalar checker, image by image."""
    gen = torch.Generator().manual_seed(404)
    B = 3
    grids = (32, 16, 8)
    ag, base = _anchors_cpu((256, 256), grids, synthetic.retinanet_anchor_sizes(3, 32))
    n = base.shape[0]
    # a different table per image: the pyramid shifted and scaled
    tables = torch.stack([base * (1.0 + 0.1 * b) + 3.0 * b for b in range(B)])
    gts = [synthetic.make_gt_boxes(m, (256, 256), gen) for m in (40, 7, 0)]
    out = kernels.match_retinanet(tables.to(device), [g.to(device) for g in gts], 0.6, 0.5, True).cpu()
    for b in range(B):
        assert np.array_equal(out[b].numpy(), c_oracle.match_retinanet(gts[b].numpy(), tables[b].numpy(), 0.6, 0.5)), b
    assert not torch.equal(out[0], kernels.match_retinanet(tables[1].to(device), [gts[0].to(device)], 0.6, 0.5, True).cpu()[0])
    # FCOS: one anchor per location
    agf, fbase = _anchors_cpu((256, 256), grids, ((8,), (16,), (32,)), (1.0,))
    ftab = torch.stack([fbase + 2.0 * b for b in range(B)])
    npl = [g * g for g in grids]
    fout = kernels.match_fcos(ftab.to(device), [g.to(device) for g in gts], npl[0], npl[-1], 1.5).cpu()
    for b in range(B):
        assert np.array_equal(fout[b].numpy(), c_oracle.match_fcos(gts[b].numpy(), ftab[b].numpy(), npl[0], npl[-1], 1.5)), b
    # candidate selection: decoded boxes depend on the image's own anchors
    C, K = 4, 150
    sizes = [9 * g * g for g in grids]
    logits = torch.empty((B, n, C)).normal_(-2.0, 1.5, generator=gen)
    reg = torch.randn((B, n, 4), generator=gen) * 0.4
    hw = torch.tensor([[256, 256], [250, 240], [256, 200]], dtype=torch.int32, device=device)
    anc_levels = [t.contiguous().to(device) for t in tables.split(sizes, dim=1)]          # [B, N_l, 4] each
    cb, cs, cl, ci, cc = kernels.select_candidates([t.to(device) for t in logits.split(sizes, dim=1)],
                                                   [t.to(device) for t in reg.split(sizes, dim=1)], anc_levels, hw, 0.3, K,
                                                   _C.CODER_BOXCODER, math.log(1000.0 / 16), want_index=True)
    cb, ci, cc = cb.cpu(), ci.cpu(), cc.cpu()
    off = 0
    for lvl, sz in enumerate(sizes):
        for b in range(B):
            s_ref, i_ref = c_oracle.filter_topk(logits[b, off:off + sz].numpy(), None, C, 0.3, K)
            cnt = int(cc[b, lvl])
            assert cnt == len(i_ref) and np.array_equal(ci[b, lvl * K:lvl * K + cnt].numpy(), i_ref), (lvl, b)
            a_idx = i_ref // C
            bx = c_oracle.decode_single(reg[b, off:off + sz].numpy()[a_idx], tables[b, off:off + sz].numpy()[a_idx],
                                        math.log(1000.0 / 16))
            bx = c_oracle.clip(bx, float(hw[b, 0]), float(hw[b, 1]))
            assert np.array_equal(cb[b, lvl * K:lvl * K + cnt].numpy().view(np.uint32), bx.view(np.uint32)), (lvl, b)
        off += sz


def test_match_retinanet_gt_without_overlap(device):
    """A GT that overlaps no anchor has max IoU 0: torchvision then restores every anchor."""
    ag, anchors = _anchors_cpu((128, 128), (16, 8), ((16,), (32,)), (1.0,))
    gt = torch.tensor([[10., 10., 60., 60.], [5000., 5000., 5010., 5010.]])
    out = kernels.match_retinanet(anchors.to(device), [gt.to(device)], 0.6, 0.5, True).cpu()[0]
    ref = tv.retinanet_match([gt], [anchors], 0.6, 0.5)[0]
    assert torch.equal(out, ref)
    assert (out >= 0).all()
    # in a batch: only the image that holds such a GT restores every anchor; with enough GT for the coarse tiles to
    # take the cooperative kernel, and an image without GT
    ag, anchors = _anchors_cpu((512, 512), (64, 32, 16, 8, 4), synthetic.retinanet_anchor_sizes(5, 32))
    gts = [synthetic.make_gt_boxes(60, (512, 512), torch.Generator().manual_seed(s)) for s in (1, 2, 3)]
    gts[1] = torch.cat((gts[1][:30], torch.tensor([[9000., 9000., 9010., 9020.]]), gts[1][30:]))
    gts.append(torch.zeros((0, 4)))
    out = kernels.match_retinanet(anchors.to(device), [g.to(device) for g in gts], 0.6, 0.5, True).cpu()
    ref = tv.retinanet_match(gts, [anchors] * len(gts), 0.6, 0.5)
    for b in range(len(gts)):
        assert torch.equal(out[b], ref[b]), f"image {b}"
    assert (out[1] >= 0).all() and (out[0] < 0).any() and (out[2] < 0).any() and (out[3] == -1).all()


def test_match_retinanet_kat(device):
    # hand-derived: IoU([0,0,10,10],[5,5,15,15]) = 25/175
    iou = kernels.box_iou(torch.tensor([[0., 0., 10., 10.]], device=device), torch.tensor([[5., 5., 15., 15.]], device=device))
    assert iou.cpu().view(torch.int32).item() == 0x3E124925
    q = torch.tensor([[.7, .55, .3, .2], [.1, .58, .45, 0.]], device=device)
    assert kernels.matcher(q, 0.6, 0.5, True).cpu().tolist() == [0, 1, -1, -1]
    assert kernels.matcher(q, 0.6, 0.5, False).cpu().tolist() == [0, -2, -1, -1]


def test_matcher_matrix_random(device):
    gen = torch.Generator().manual_seed(5)
    q = torch.rand((37, 5000), generator=gen)
    q[3, 100] = q[3].max()  # tie on a row maximum
    out = kernels.matcher(q.to(device), 0.6, 0.5, True).cpu()
    assert torch.equal(out, tv.Matcher(0.6, 0.5, True)(q.clone()))
    a = torch.rand((50, 4), generator=gen) * 100
    a[:, 2:] += a[:, :2]
    b = torch.rand((3000, 4), generator=gen) * 100
    b[:, 2:] += b[:, :2]
    assert torch.equal(kernels.box_iou(a.to(device), b.to(device)).cpu(), tv.box_iou(a, b))


@pytest.mark.parametrize("m", [1, 33, 200, 1500])
def test_match_fcos(device, m):
    grids = (64, 32, 16, 8, 4)
    ag, anchors = _anchors_cpu((512, 512), grids, ((8,), (16,), (32,), (64,), (128,)), (1.0,))
    npl = [g * g for g in grids]
    gts = [synthetic.make_gt_boxes(m, (512, 512), torch.Generator().manual_seed(7 * m + b)) for b in range(2)]
    gts.append(torch.zeros((0, 4)))
    out = kernels.match_fcos(anchors.to(device), [g.to(device) for g in gts], npl[0], npl[-1], 1.5).cpu()
    ref = tv.fcos_match(gts, [anchors] * 3, npl, 1.5)
    for b in range(3):
        assert torch.equal(out[b], ref[b]), f"image {b}"
        assert np.array_equal(out[b].numpy(), c_oracle.match_fcos(gts[b].numpy(), anchors.numpy(), npl[0], npl[-1], 1.5))
    assert (out[0] >= 0).any()
    # a box whose area exceeds 1e8 turns the value of a match negative (1e8 - area) and that of a miss into -0.0:
    # first and in the middle of the list
    huge = torch.tensor([[-9744., -9744., 10256., 10256.]])
    for pos in (0, min(3, m)):
        g = torch.cat((gts[0][:pos], huge, gts[0][pos:]))
        out = kernels.match_fcos(anchors.to(device), [g.to(device)], npl[0], npl[-1], 1.5).cpu()
        assert torch.equal(out[0], tv.fcos_match([g], [anchors], npl, 1.5)[0]), f"huge box at {pos}"


def test_decode_boxes(device):
    gen = torch.Generator().manual_seed(11)
    anchors = torch.rand((4097, 4), generator=gen) * 500
    anchors[:, 2:] += anchors[:, :2] + 1
    rel = torch.randn((4097, 4), generator=gen)
    rel[5, 2] = 5.0
    rel[6, 3] = 100.0
    rel[7, 2] = -120.0
    clip = math.log(1000.0 / 16)
    out = kernels.decode_boxes(rel.to(device), anchors.to(device), _C.CODER_BOXCODER, clip).cpu()
    ref = c_oracle.decode_single(rel.numpy(), anchors.numpy(), clip)
    assert np.array_equal(out.numpy().view(np.uint32), ref.view(np.uint32))
    ref_spec = tv.BoxCoder().decode_single(rel, anchors, math="spec")
    assert torch.equal(out, ref_spec)
    ref_torch = tv.BoxCoder().decode_single(rel, anchors, math="torch")
    assert torch.allclose(out, ref_torch, rtol=1e-5, atol=1e-4)
    out_c = kernels.decode_boxes(rel.to(device), anchors.to(device), _C.CODER_BOXCODER, clip, (300, 400)).cpu()
    assert torch.equal(out_c, tv.clip_boxes_to_image(ref_spec, (300, 400)))
    out_l = kernels.decode_boxes(rel.to(device), anchors.to(device), _C.CODER_LINEAR, 0.0).cpu()
    assert torch.equal(out_l, tv.BoxLinearCoder(True).decode(rel, anchors))


def _random_boxes(n, gen, extent=400.0, wh=80.0):
    xy = torch.rand((n, 2), generator=gen) * extent
    sz = torch.rand((n, 2), generator=gen) * wh + 1
    return torch.cat([xy, xy + sz], dim=1)


@pytest.mark.parametrize("n,classes,extent", [(1, 1, 400), (65, 3, 100), (300, 1, 60), (1000, 4, 400), (1500, 5, 150), (6000, 4, 800),
                                              (9000, 2, 500)])
def test_batched_nms(device, n, classes, extent):
    gen = torch.Generator().manual_seed(n + classes)
    boxes = _random_boxes(n, gen, float(extent))
    scores = torch.rand(n, generator=gen)
    scores[::7] = scores[0]        # ties in score
    labels = torch.randint(0, classes, (n,), generator=gen)
    cnt = torch.tensor([[n]], dtype=torch.int32, device=device)
    for mode, omode in ((_C.NMS_TRICK, 0), (_C.NMS_VANILLA, 1)):
        for thr in (0.3, 0.5):
            ref = c_oracle.batched_nms(boxes.numpy(), scores.numpy(), labels.numpy(), thr, omode)
            keep, ob, os_, ol, oc = kernels.batched_nms_padded(boxes[None].to(device), scores[None].to(device),
                                                               labels[None].to(device), cnt, n, False, thr, mode, n)
            k = int(oc.item())
            assert k == len(ref), (mode, thr, k, len(ref))
            assert np.array_equal(keep[0, :k].cpu().numpy(), ref), (mode, thr)
            assert torch.equal(ob[0, :k].cpu(), boxes[ref])
            assert torch.equal(os_[0, :k].cpu(), scores[ref])
            assert torch.equal(ol[0, :k].cpu(), labels[ref])
            # capped: first D of the same list
            d = max(1, min(50, len(ref) // 2))
            keep2, _, _, _, oc2 = kernels.batched_nms_padded(boxes[None].to(device), scores[None].to(device),
                                                            labels[None].to(device), cnt, n, False, thr, mode, d)
            assert int(oc2.item()) == min(d, len(ref))
            assert np.array_equal(keep2[0, :int(oc2.item())].cpu().numpy(), ref[:d])
    # torch restatement agrees with the C restatement on this input
    t_keep = tv.batched_nms(boxes, scores, labels, 0.5, "cpu")
    o_keep = c_oracle.batched_nms(boxes.numpy(), scores.numpy(), labels.numpy(), 0.5, 1 if 4 * n > 4000 else 0)
    assert np.array_equal(t_keep.numpy(), o_keep)


def _segmented_nms_case(gen, batch, num_seg, seg_stride, classes, extent):
    """Padded candidate arrays as the selection kernel hands them over: every segment sorted by (score desc, index
    asc), ragged counts, one empty segment, ties in score inside and across segments."""
    n = num_seg * seg_stride
    boxes = torch.stack([_random_boxes(n, gen, float(extent)) for _ in range(batch)])
    scores = torch.rand((batch, n), generator=gen)
    scores[:, ::7] = 0.75
    scores[:, 1::11] = scores[:, :1]
    scores = scores.view(batch, num_seg, seg_stride).sort(dim=2, descending=True, stable=True)[0].reshape(batch, n)
    labels = torch.randint(0, classes, (batch, n), generator=gen, dtype=torch.int32)
    counts = torch.randint(seg_stride // 3, seg_stride + 1, (batch, num_seg), generator=gen, dtype=torch.int32)
    counts[0, 0] = seg_stride
    counts[batch - 1, num_seg - 1] = 0
    return boxes, scores, labels, counts


@pytest.mark.parametrize("num_seg,seg_stride,classes,max_keep,extent", [
    (6, 1000, 4, 500, 1024),          # the benchmark's shape
    (6, 5000, 91, 300, 1536),         # topk_candidates = 5000: 30 000 candidates per image, lazily merged
    (6, 5000, 91, 3000, 1536),        # kept list beyond the LDS cache, many classes
    (8, 700, 1, 2000, 300),           # one class, dense boxes: long kept list, many strips
    (3, 64, 7, 64, 100),
    (12, 300, 5, 200, 400),           # more segments than are merged lazily: ranked first
])
def test_batched_nms_sorted_segments(device, num_seg, seg_stride, classes, max_keep, extent):
    """od_batched_nms on pre-sorted padded segments (what postprocess hands over) against the scalar checker run on
    the concatenation of the valid entries: keep indices, boxes, scores, labels and counts, both branch semantics,
    both threshold rules."""
    gen = torch.Generator().manual_seed(num_seg * 131 + seg_stride + classes)
    B = 2
    boxes, scores, labels, counts = _segmented_nms_case(gen, B, num_seg, seg_stride, classes, extent)
    dev = [t.to(device) for t in (boxes, scores, labels, counts)]
    for mode, omode in ((_C.NMS_TRICK, 0), (_C.NMS_VANILLA, 1)):
        for flag, frule, thr in ((0, False, 0.3), (_C.NMS_THRESHOLD_FLOAT, True, 0.5)):
            keep, ob, os_, ol, oc = kernels.batched_nms_padded(dev[0], dev[1], dev[2], dev[3], seg_stride, True, thr,
                                                               mode | flag, max_keep)
            for b in range(B):
                valid = (torch.arange(seg_stride)[None, :] < counts[b][:, None]).reshape(-1)
                vb, vs, vl = boxes[b][valid], scores[b][valid], labels[b][valid]
                ref = c_oracle.batched_nms(vb.numpy(), vs.numpy(), vl.numpy(), thr, omode, float_rule=frule)[:max_keep]
                k = int(oc[b].item())
                assert k == len(ref), (mode, thr, b, k, len(ref))
                assert np.array_equal(keep[b, :k].cpu().numpy(), ref), (mode, thr, b)
                assert torch.equal(ob[b, :k].cpu(), vb[ref]) and torch.equal(os_[b, :k].cpu(), vs[ref])
                assert torch.equal(ol[b, :k].cpu(), vl[ref].long())
                assert float(ob[b, k:].abs().sum()) == 0.0 and bool((keep[b, k:] == -1).all())


def test_batched_nms_large_unsorted_many_classes(device):
    """30 000 unsorted candidates with 91 classes in one segment (ranked by the ordering kernel), both branches."""
    gen = torch.Generator().manual_seed(30091)
    n, classes = 30000, 91
    boxes = _random_boxes(n, gen, 1500.0)
    scores = torch.rand(n, generator=gen)
    scores[::9] = scores[3]
    labels = torch.randint(0, classes, (n,), generator=gen)
    cnt = torch.tensor([[n]], dtype=torch.int32, device=device)
    for mode, omode in ((_C.NMS_TRICK, 0), (_C.NMS_VANILLA, 1)):
        ref = c_oracle.batched_nms(boxes.numpy(), scores.numpy(), labels.numpy(), 0.5, omode)
        for max_keep in (100, n):
            keep, ob, os_, ol, oc = kernels.batched_nms_padded(boxes[None].to(device), scores[None].to(device),
                                                               labels[None].to(device), cnt, n, False, 0.5, mode, max_keep)
            k = int(oc.item())
            assert k == min(max_keep, len(ref)), (mode, max_keep, k, len(ref))
            assert np.array_equal(keep[0, :k].cpu().numpy(), ref[:k]), (mode, max_keep)
            assert torch.equal(ob[0, :k].cpu(), boxes[ref[:k]]) and torch.equal(ol[0, :k].cpu(), labels[ref[:k]])


def test_batched_nms_trick_when_classes_are_not_separated(device):
    """The coordinate trick separates the classes only for non-negative coordinates and labels and a threshold >= 0;
    otherwise boxes of different classes do interact in torchvision's single nms call, and must here."""
    gen = torch.Generator().manual_seed(99)
    n = 700
    boxes = _random_boxes(n, gen, 300.0, 120.0) - 150.0                 # coordinates in [-150, 270]
    scores = torch.rand(n, generator=gen)
    labels = torch.randint(0, 3, (n,), generator=gen)
    cnt = torch.tensor([[n]], dtype=torch.int32, device=device)
    cases = [(boxes, labels, 0.3), (boxes + 150.0, labels, -0.5), (boxes + 150.0, labels - 1, 0.3),
             (boxes + 150.0, labels * 40000, 0.3)]                       # label * (max + 1) beyond 2^23
    for bx, lab, thr in cases:
        ref = c_oracle.batched_nms(bx.numpy(), scores.numpy(), lab.numpy(), thr, 0)
        keep, _, _, ol, oc = kernels.batched_nms_padded(bx[None].to(device), scores[None].to(device),
                                                        lab[None].to(device), cnt, n, False, thr, _C.NMS_TRICK, n)
        k = int(oc.item())
        assert k == len(ref), (thr, k, len(ref))
        assert np.array_equal(keep[0, :k].cpu().numpy(), ref), thr
        assert torch.equal(ol[0, :k].cpu(), lab[ref])
    # per-class branch with labels too far apart to sort (class, rank) as one 32-bit key: candidates stay ungrouped
    wide = labels * (1 << 29) - 5
    ref = c_oracle.batched_nms((boxes + 150.0).numpy(), scores.numpy(), wide.numpy(), 0.3, 1)
    keep, _, _, ol, oc = kernels.batched_nms_padded((boxes + 150.0)[None].to(device), scores[None].to(device),
                                                    wide[None].to(device), cnt, n, False, 0.3, _C.NMS_VANILLA, n)
    k = int(oc.item())
    assert k == len(ref) and np.array_equal(keep[0, :k].cpu().numpy(), ref)
    assert torch.equal(ol[0, :k].cpu(), wide[ref])
    # separated classes in the trick branch differ from nothing: the vanilla branch gives the same keep list there
    ref_v = c_oracle.batched_nms((boxes + 150.0).numpy(), scores.numpy(), labels.numpy(), 0.3, 1)
    ref_t = c_oracle.batched_nms((boxes + 150.0).numpy(), scores.numpy(), labels.numpy(), 0.3, 0)
    assert len(ref_v) > 0 and len(ref_t) > 0


@pytest.mark.parametrize("n,classes,extent", [(1, 1, 400), (65, 3, 100), (300, 1, 60), (1500, 5, 150), (4145, 4, 1024), (6000, 1, 800),
                                              (8192, 7, 500)])
def test_batched_nms_matrix_form(device, n, classes, extent):
    """OD_NMS_PREFER_MATRIX: the chip-wide bit-matrix form gives the checker's keep list (both branches, both
    threshold rules, score ties, caps), i.e. what the strip kernel gives; unsorted input in one segment."""
    gen = torch.Generator().manual_seed(7 * n + classes)
    boxes = _random_boxes(n, gen, float(extent))
    scores = torch.rand(n, generator=gen)
    scores[::7] = scores[0]
    labels = torch.randint(0, classes, (n,), generator=gen)
    cnt = torch.tensor([[n]], dtype=torch.int32, device=device)
    M = _C.NMS_PREFER_MATRIX
    assert _C.lib().od_batched_nms_matrix_workspace_bytes(1, 1, n) > _C.lib().od_batched_nms_workspace_bytes(1, 1, n)
    dev = (boxes[None].to(device), scores[None].to(device), labels[None].to(device))
    for mode, omode in ((_C.NMS_TRICK, 0), (_C.NMS_VANILLA, 1)):
        for flag, frule, thr in ((0, False, 0.3), (_C.NMS_THRESHOLD_FLOAT, True, 0.5)):
            ref = c_oracle.batched_nms(boxes.numpy(), scores.numpy(), labels.numpy(), thr, omode, float_rule=frule)
            for max_keep in (n, max(1, len(ref) // 3)):
                keep, ob, os_, ol, oc = kernels.batched_nms_padded(*dev, cnt, n, False, thr, mode | flag | M, max_keep)
                k = int(oc.item())
                assert k == min(max_keep, len(ref)), (mode, thr, k, len(ref))
                assert np.array_equal(keep[0, :k].cpu().numpy(), ref[:k]), (mode, thr)
                assert torch.equal(ob[0, :k].cpu(), boxes[ref[:k]]) and torch.equal(os_[0, :k].cpu(), scores[ref[:k]])
                assert torch.equal(ol[0, :k].cpu(), labels[ref[:k]])
                assert float(ob[0, k:].abs().sum()) == 0.0 and bool((keep[0, k:] == -1).all())
    # no labels (plain nms) through the public wrapper, which picks this form above 1024 boxes
    from object_detection_amd import boxes as box_ops
    ref = c_oracle.nms(boxes.numpy(), scores.numpy(), 0.4, True)
    assert np.array_equal(box_ops.nms(boxes.to(device), scores.to(device), 0.4).cpu().numpy(), ref)


def test_batched_nms_matrix_form_segments_and_mixed_classes(device):
    """The matrix form on what the pipeline hands over (sorted padded segments, ragged counts, a batch) and on inputs
    whose classes the coordinate trick does not separate (negative coordinates, negative threshold, negative labels,
    offsets beyond 2^23); too many boxes for the form: the flag is ignored."""
    gen = torch.Generator().manual_seed(4242)
    B, num_seg, seg_stride, max_keep = 3, 6, 1000, 500
    boxes, scores, labels, counts = _segmented_nms_case(gen, B, num_seg, seg_stride, 4, 1024)
    dev = [t.to(device) for t in (boxes, scores, labels, counts)]
    for mode, omode in ((_C.NMS_AUTO_GPU, 0), (_C.NMS_VANILLA, 1)):
        keep, ob, os_, ol, oc = kernels.batched_nms_padded(dev[0], dev[1], dev[2], dev[3], seg_stride, True, 0.3,
                                                           mode | _C.NMS_PREFER_MATRIX | _C.NMS_THRESHOLD_FLOAT, max_keep)
        plain = kernels.batched_nms_padded(dev[0], dev[1], dev[2], dev[3], seg_stride, True, 0.3,
                                           mode | _C.NMS_THRESHOLD_FLOAT, max_keep)
        for b in range(B):
            valid = (torch.arange(seg_stride)[None, :] < counts[b][:, None]).reshape(-1)
            vb, vs, vl = boxes[b][valid], scores[b][valid], labels[b][valid]
            ref = c_oracle.batched_nms(vb.numpy(), vs.numpy(), vl.numpy(), 0.3, omode, float_rule=True)[:max_keep]
            k = int(oc[b].item())
            assert k == len(ref) and np.array_equal(keep[b, :k].cpu().numpy(), ref), (mode, b)
            assert torch.equal(ob[b, :k].cpu(), vb[ref]) and torch.equal(ol[b, :k].cpu(), vl[ref].long())
        for got, want in zip((keep, ob, os_, ol, oc), plain):
            assert torch.equal(got, want)
    n = 700
    bx0 = _random_boxes(n, gen, 300.0, 120.0) - 150.0
    sc = torch.rand(n, generator=gen)
    lab0 = torch.randint(0, 3, (n,), generator=gen)
    cnt = torch.tensor([[n]], dtype=torch.int32, device=device)
    for bx, lab, thr in [(bx0, lab0, 0.3), (bx0 + 150.0, lab0, -0.5), (bx0 + 150.0, lab0 - 1, 0.3), (bx0 + 150.0, lab0 * 40000, 0.3)]:
        ref = c_oracle.batched_nms(bx.numpy(), sc.numpy(), lab.numpy(), thr, 0)
        keep, _, _, ol, oc = kernels.batched_nms_padded(bx[None].to(device), sc[None].to(device), lab[None].to(device), cnt, n,
                                                        False, thr, _C.NMS_TRICK | _C.NMS_PREFER_MATRIX, n)
        k = int(oc.item())
        assert k == len(ref) and np.array_equal(keep[0, :k].cpu().numpy(), ref), thr
        assert torch.equal(ol[0, :k].cpu(), lab[ref])
    # 9000 boxes: beyond the form's limit, the strip kernel answers
    assert _C.lib().od_batched_nms_matrix_workspace_bytes(1, 1, 9000) == 0
    n = 9000
    bx, sc2 = _random_boxes(n, gen, 500.0), torch.rand(n, generator=gen)
    ref = c_oracle.nms(bx.numpy(), sc2.numpy(), 0.5, True)
    cnt = torch.tensor([[n]], dtype=torch.int32, device=device)
    keep, _, _, _, oc = kernels.batched_nms_padded(bx[None].to(device), sc2[None].to(device), None, cnt, n, False, 0.5,
                                                   _C.NMS_TRICK | _C.NMS_THRESHOLD_FLOAT | _C.NMS_PREFER_MATRIX, n, want_labels=False)
    assert np.array_equal(keep[0, :int(oc.item())].cpu().numpy(), ref)


def test_batched_nms_matrix_form_edge_cases(device):
    """The matrix form on a batch with an image without candidates, without labels, with max_keep 1 and with a cap
    that ends the walk inside a block."""
    gen = torch.Generator().manual_seed(808)
    B, n = 3, 700
    boxes = torch.stack([_random_boxes(n, gen, 200.0) for _ in range(B)])
    scores = torch.rand((B, n), generator=gen)
    counts = torch.tensor([[n], [0], [333]], dtype=torch.int32)
    dev = [t.to(device) for t in (boxes, scores, counts)]
    flags = _C.NMS_TRICK | _C.NMS_THRESHOLD_FLOAT | _C.NMS_PREFER_MATRIX
    for max_keep in (1, 70, n):
        keep, ob, os_, _, oc = kernels.batched_nms_padded(dev[0], dev[1], None, dev[2], n, False, 0.4, flags, max_keep,
                                                          want_labels=False)
        for b in range(B):
            m = int(counts[b, 0])
            ref = c_oracle.nms(boxes[b, :m].numpy(), scores[b, :m].numpy(), 0.4, True)[:max_keep] if m else np.zeros(0, np.int64)
            k = int(oc[b].item())
            assert k == len(ref), (max_keep, b, k, len(ref))
            assert np.array_equal(keep[b, :k].cpu().numpy(), ref), (max_keep, b)
            assert torch.equal(ob[b, :k].cpu(), boxes[b][ref]) and torch.equal(os_[b, :k].cpu(), scores[b][ref])
            assert bool((keep[b, k:] == -1).all()) and float(ob[b, k:].abs().sum()) == 0.0


def test_nms_kat_and_degenerate(device):
    b = torch.tensor([[0., 0., 10., 10.], [1., 1., 11., 11.], [20., 20., 30., 30.]])
    s = torch.tensor([.9, .8, .7])
    cnt = torch.tensor([[3]], dtype=torch.int32, device=device)
    keep, _, _, _, oc = kernels.batched_nms_padded(b[None].to(device), s[None].to(device), None, cnt, 3, False, 0.5,
                                                   _C.NMS_TRICK, 3)
    assert keep[0, :int(oc.item())].cpu().tolist() == [0, 2]
    # zero-area pair: IoU = 0/0 = NaN never suppresses
    z = torch.tensor([[5., 5., 5., 5.], [5., 5., 5., 5.]])
    cnt2 = torch.tensor([[2]], dtype=torch.int32, device=device)
    keep, _, _, _, oc = kernels.batched_nms_padded(z[None].to(device), torch.tensor([[.5, .4]], device=device), None,
                                                   cnt2, 2, False, 0.5, _C.NMS_TRICK, 2)
    assert keep[0, :int(oc.item())].cpu().tolist() == [0, 1]
    # IoU exactly float(thr): CPU semantics compare the float IoU against the DOUBLE threshold -> suppressed
    # boxes [0,0,10,10] and [0,0,10,3]: inter 30, union 100 -> 0.3f > 0.3 (double)
    e = torch.tensor([[0., 0., 10., 10.], [0., 0., 10., 3.]])
    keep, _, _, _, oc = kernels.batched_nms_padded(e[None].to(device), torch.tensor([[.9, .8]], device=device), None,
                                                   cnt2, 2, False, 0.3, _C.NMS_TRICK, 2)
    assert keep[0, :int(oc.item())].cpu().tolist() == c_oracle.nms(e.numpy(), np.float32([.9, .8]), 0.3).tolist() == [0]
    # the CUDA kernel's rule compares the float IoU with float(0.3): 0.3f > 0.3f is false -> both kept
    keep, _, _, _, oc = kernels.batched_nms_padded(e[None].to(device), torch.tensor([[.9, .8]], device=device), None,
                                                   cnt2, 2, False, 0.3, _C.NMS_TRICK | _C.NMS_THRESHOLD_FLOAT, 2)
    assert keep[0, :int(oc.item())].cpu().tolist() == [0, 1]
    assert c_oracle.nms(e.numpy(), np.float32([.9, .8]), 0.3, float_rule=True).tolist() == [0, 1]
    assert tv.nms(e, torch.tensor([.9, .8]), 0.3, float_rule=True).tolist() == [0, 1]
    assert tv.nms(e, torch.tensor([.9, .8]), 0.3).tolist() == [0]
    from object_detection_amd import boxes as box_ops
    sc = torch.tensor([.9, .8], device=device)
    assert box_ops.nms(e.to(device), sc, 0.3).cpu().tolist() == [0, 1]
    assert box_ops.nms(e.to(device), sc, 0.3, threshold_rule="cpu").cpu().tolist() == [0]
    # empty image
    cnt0 = torch.tensor([[0]], dtype=torch.int32, device=device)
    _, _, _, _, oc = kernels.batched_nms_padded(e[None].to(device), torch.tensor([[.9, .8]], device=device), None,
                                                cnt0, 2, False, 0.3, _C.NMS_TRICK, 2)
    assert int(oc.item()) == 0


@pytest.mark.parametrize("coder", ["retinanet", "fcos"])
def test_select_candidates(device, coder):
    cfg = synthetic.CFG1
    head = synthetic.make_head_outputs(cfg, with_ctrness=(coder == "fcos"))
    sizes = synthetic.level_sizes(cfg)
    ag, anchors = _anchors_cpu(cfg.image_size, cfg.grids, synthetic.retinanet_anchor_sizes(5, 32))
    anchors_l = list(anchors.split(sizes))
    cls_l = list(head["cls_logits"].split(sizes, dim=1))
    reg_l = list(head["bbox_regression"].split(sizes, dim=1))
    ctr_l = list(head["bbox_ctrness"].split(sizes, dim=1)) if coder == "fcos" else None
    hw = torch.tensor([[512, 512], [480, 500]], dtype=torch.int32, device=device)
    K, thr = 1000, (0.5 if coder == "retinanet" else 0.2)
    code = _C.CODER_BOXCODER if coder == "retinanet" else _C.CODER_LINEAR
    dev = lambda ts: [t.to(device) for t in ts]
    cls_d = head["cls_logits"].to(device)
    reg_d = head["bbox_regression"].to(device)
    cb, cs, cl, ci, cc = kernels.select_candidates(
        list(cls_d.split(sizes, dim=1)), list(reg_d.split(sizes, dim=1)), dev(anchors_l), hw, thr, K, code,
        math.log(1000.0 / 16), ctr_levels=dev(ctr_l) if ctr_l else None, want_index=True)
    cb, cs, cl, ci, cc = cb.cpu(), cs.cpu(), cl.cpu(), ci.cpu(), cc.cpu()
    shapes = [(512, 512), (480, 500)]
    for b in range(cfg.batch):
        for l in range(len(sizes)):
            s_ref, i_ref = c_oracle.filter_topk(cls_l[l][b].numpy(), ctr_l[l][b].numpy() if ctr_l else None,
                                                cfg.num_classes, thr, K)
            n = int(cc[b, l])
            assert n == len(s_ref), (b, l, n, len(s_ref))
            sl = slice(l * K, l * K + n)
            assert np.array_equal(ci[b, sl].numpy(), i_ref), (b, l)
            assert np.array_equal(cs[b, sl].numpy().view(np.uint32), s_ref.view(np.uint32))
            a_idx = torch.from_numpy(i_ref // cfg.num_classes)
            assert np.array_equal(cl[b, sl].numpy(), i_ref % cfg.num_classes)
            if coder == "retinanet":
                bx = tv.BoxCoder().decode_single(reg_l[l][b][a_idx], anchors_l[l][a_idx], math="spec")
            else:
                bx = tv.BoxLinearCoder(True).decode(reg_l[l][b][a_idx], anchors_l[l][a_idx])
            bx = tv.clip_boxes_to_image(bx, shapes[b])
            assert torch.equal(cb[b, sl], bx), (b, l)
    assert int(cc[:, 0].min()) == K   # the finest level saturates top-k (ties at score 1.0 included)


def _check_select_single_level(device, logits, C, thr, K, ctrness=None):
    """One level, batch of logits [B, N, C] (and FCOS centre-ness [B, N, 1]): indices and scores of the top-k against
    the scalar checker."""
    B, N = logits.shape[0], logits.shape[1]
    reg = torch.zeros((B, N, 4))
    anchors = torch.tensor([[0., 0., 8., 8.]]).repeat(N, 1)
    hw = torch.tensor([[64, 64]] * B, dtype=torch.int32, device=device)
    cb, cs, cl, ci, cc = kernels.select_candidates([logits.to(device)], [reg.to(device)], [anchors.to(device)], hw, thr, K,
                                                   _C.CODER_BOXCODER if ctrness is None else _C.CODER_LINEAR,
                                                   math.log(1000.0 / 16), want_index=True,
                                                   ctr_levels=None if ctrness is None else [ctrness.to(device)])
    cs, ci, cc = cs.cpu(), ci.cpu(), cc.cpu()
    for b in range(B):
        s_ref, i_ref = c_oracle.filter_topk(logits[b].numpy(), None if ctrness is None else ctrness[b].numpy().reshape(-1),
                                            C, thr, K)
        n = int(cc[b, 0])
        assert n == len(s_ref), (b, n, len(s_ref))
        assert np.array_equal(ci[b, :n].numpy(), i_ref), b
        assert np.array_equal(cs[b, :n].numpy().view(np.uint32), s_ref.view(np.uint32)), b


def test_select_ties_beyond_lds_capacity(device):
    """More candidates than fit in LDS, and the K-th best lies inside a run of thousands of equal scores:
    the radix descent has to go all the way down to the index bits (tie rule: lower flat index first)."""
    gen = torch.Generator().manual_seed(7)
    N, C = 30011, 3                                      # N*C is odd: exercises the unaligned chunk tail
    logits = torch.empty((2, N, C)).normal_(-1.0, 1.0, generator=gen)
    tie = torch.rand((2, N, C), generator=gen) < 0.3     # ~27k logits with one identical value above the rest
    logits[tie] = 6.0
    logits[1][torch.rand((N, C), generator=gen) < 0.001] = 30.0   # a few saturated scores on top (image 1)
    for K in (1000, 4096):
        _check_select_single_level(device, logits, C, 0.5, K)


def test_select_tied_groups_in_lds(device):
    """Fewer candidates than LDS holds, in groups of equal scores: a pass of the radix descent that meets keys which
    agree in the bits it looks at skips to the highest bit that varies -- score bits when two groups differ in the last
    bits of the score, index bits inside one group.  K inside the first group, inside the second, at a group's edge,
    and above / below the thread count of the top-k workgroup (1024: one key per thread or several)."""
    gen = torch.Generator().manual_seed(13)
    N, C = 2500, 3                                       # 7500 logits
    logits = torch.empty((2, N, C)).normal_(-2.0, 1.0, generator=gen)
    u = torch.rand((2, N, C), generator=gen)
    a = 3.0
    logits[u < 0.30] = a                                 # ~2250 equal scores on top ...
    logits[(u >= 0.30) & (u < 0.55)] = float(np.nextafter(np.float32(a), np.float32(0)))   # ... ~1900 one ulp below
    logits[1][(u[1] >= 0.55) & (u[1] < 0.60)] = 40.0     # image 1: ~375 saturated scores above both groups
    for K in (100, 1000, 1024, 1025, 2250, 3000, 4500):
        _check_select_single_level(device, logits, C, 0.5, K)
    # the same with centre-ness (scores are products, the groups stay groups)
    ctr = torch.full((2, N, 1), 2.0)
    _check_select_single_level(device, logits, C, 0.2, 1000, ctrness=ctr)


def test_select_all_equal_and_no_threshold(device):
    """Every score identical (order = flat index), and score_thresh <= 0 (nothing is prefiltered)."""
    logits = torch.full((1, 5000, 2), 0.25)
    _check_select_single_level(device, logits, 2, 0.05, 1000)      # 10000 equal candidates > LDS capacity
    _check_select_single_level(device, logits[:, :3000], 2, 0.05, 1000)   # 6000 equal candidates, all in LDS
    gen = torch.Generator().manual_seed(11)
    small = torch.empty((2, 777, 3)).normal_(0.0, 3.0, generator=gen)
    _check_select_single_level(device, small, 3, 0.0, 300)
    _check_select_single_level(device, small, 3, -1.0, 4096)   # every element passes, K above the count


def test_match_retinanet_heavy_entry_overflow(device):
    """GT boxes that cover almost the whole image meet every anchor tile: the per-GT entry budget of the
    cooperative path overflows and the restore step falls back to re-examining the tiles."""
    cfg = synthetic.CFG2
    ag, anchors = _anchors_cpu(cfg.image_size, cfg.grids[1:], synthetic.retinanet_anchor_sizes(6)[1:])   # 196,560 anchors
    gen = torch.Generator().manual_seed(3)
    m = 30
    lo = torch.rand((m, 2), generator=gen) * 40.0
    hi = 1024.0 - torch.rand((m, 2), generator=gen) * 40.0
    big = torch.cat([lo, hi], dim=1)
    small = synthetic.make_gt_boxes(10, cfg.image_size, gen)
    gts = [torch.cat([big, small]), big[:26].clone()]
    out = kernels.match_retinanet(anchors.to(device), [g.to(device) for g in gts], 0.6, 0.5, True).cpu()
    for b, g in enumerate(gts):
        ref = c_oracle.match_retinanet(g.numpy(), anchors.numpy(), 0.6, 0.5, True)
        assert np.array_equal(out[b].numpy(), ref), f"image {b}"


def test_select_prefilter_margin_leaves_empty_slots(device):
    """Logits inside the prefilter margin (just below logit(thresh)) are queued but fail the exact threshold:
    their reserved slots hold zero keys, which the top-k must ignore -- also when fewer than K real keys
    remain among thousands of empty slots."""
    gen = torch.Generator().manual_seed(5)
    # thresh 0.5 -> logit 0; the prefilter bound is -1e-3: values in (-1e-3, 0] are queued and then rejected
    band = lambda shape: -torch.rand(shape, generator=gen) * 0.9e-3
    mix = torch.empty((2, 30000, 3)).uniform_(-0.002, 0.002, generator=gen)          # ~45k pass, ~20k empty slots
    _check_select_single_level(device, mix, 3, 0.5, 1000)
    few = band((2, 1800, 3))                                                          # 5400 empty slots (fit in LDS) ...
    few[0].view(-1)[torch.randperm(5400, generator=gen)[:300]] = 1.5                  # ... and 300 real keys < K
    _check_select_single_level(device, few, 3, 0.5, 1000)
    many = band((2, 7000, 3))                                                         # 21000 empty slots (beyond LDS)
    many[1].view(-1)[torch.randperm(21000, generator=gen)[:700]] = 2.0
    _check_select_single_level(device, many, 3, 0.5, 1000)


def test_match_retinanet_dense_targets(device):
    """2000 GT boxes per image (the dense-matching configuration, on the small pyramid): the GT list is streamed in
    chunks instead of staying in registers, and almost every tile goes through the cooperative kernel."""
    grids = (64, 32, 16, 8, 4)
    ag, anchors = _anchors_cpu((512, 512), grids, synthetic.retinanet_anchor_sizes(5, 32))
    gts = []
    for b, m in enumerate((2000, 1300)):
        gen = torch.Generator().manual_seed(900 + b)
        gts.append(synthetic.make_gt_boxes(m, (512, 512), gen))
    out = kernels.match_retinanet(anchors.to(device), [g.to(device) for g in gts], 0.6, 0.5, True).cpu()
    for b, g in enumerate(gts):
        ref = c_oracle.match_retinanet(g.numpy(), anchors.numpy(), 0.6, 0.5, True)
        assert np.array_equal(out[b].numpy(), ref), f"image {b}"


def test_select_many_classes(device):
    """91 classes (flat index = anchor * 91 + class), low score threshold, K above and below the candidate count."""
    gen = torch.Generator().manual_seed(21)
    logits = torch.empty((2, 3000, 91)).normal_(-4.0, 1.5, generator=gen)
    for thr, K in ((0.05, 1000), (0.3, 4096)):
        _check_select_single_level(device, logits, 91, thr, K)


@pytest.mark.parametrize("num_classes,n,pad", [(1, 5000, 0), (3, 4099, 5), (4, 8192, 0), (4, 3000, 3), (6, 7001, 0),
                                               (91, 1234, 7), (80, 2048, 4)])
def test_focal_loss_kernels_class_counts_and_ragged_sizes(device, num_classes, n, pad):
    """Fused classification loss for class counts below four (general path), multiples of four (a group of four
    logits stays inside one anchor) and others (groups straddle anchors), sizes that leave a partial last chunk and
    image strides that break 16-byte alignment.  Checked against sigmoid_focal_loss evaluated in fp64 by torch:
    loss 1e-5 relative, gradient 1e-5 of its largest entry plus 1e-4 relative per element."""
    g = torch.Generator().manual_seed(1000 + num_classes * 7 + n)
    B, M = 3, 9
    storage = torch.randn((B, n * num_classes + pad), generator=g) * 2.0 - 2.0
    logits = storage[:, :n * num_classes].view(B, n, num_classes)
    matched = torch.randint(-2, M, (B, n), generator=g)
    matched[1] = torch.where(matched[1] >= 0, torch.full_like(matched[1], -1), matched[1])      # image without foreground
    labels = torch.randint(0, num_classes, (B * M,), generator=g)
    offs = [0, M, 2 * M, 3 * M]
    for alpha, gamma in ((0.25, 2.0), (-1.0, 2.0), (0.6, 0.5)):
        x = logits.double().requires_grad_(True)
        total = 0.0
        for b in range(B):
            t = torch.zeros((n, num_classes), dtype=torch.float64)
            fg = matched[b] >= 0
            t[fg, labels[offs[b]:offs[b + 1]][matched[b][fg]]] = 1.0
            valid = matched[b] != -2
            total = total + tv.sigmoid_focal_loss(x[b][valid], t[valid], alpha=alpha, gamma=gamma,
                                                  reduction="sum") / max(1, int(fg.sum()))
        ref = total / B
        ref.backward()
        xd = storage.to(device)[:, :n * num_classes].view(B, n, num_classes)
        assert xd.stride(0) == n * num_classes + pad
        md, ld, od = matched.to(device), labels.to(device), kernels._device_ints(offs, device)
        loss, per_image, nfg = kernels.focal_loss_forward(xd, md, ld, od, alpha, gamma)
        assert nfg.cpu().tolist() == [int((matched[b] >= 0).sum()) for b in range(B)]
        assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item()), (alpha, gamma, loss.item(), ref.item())
        grad = kernels.focal_loss_backward(xd, md, ld, od, alpha, gamma, nfg, torch.ones((1,), device=device)).cpu()
        gr = x.grad.float()
        tol = 1e-5 * gr.abs().max() + 1e-4 * gr.abs()
        assert bool(((grad - gr).abs() <= tol).all()), (alpha, gamma, float((grad - gr).abs().max()))
        assert torch.equal(grad[matched == -2], torch.zeros_like(grad[matched == -2]))


def _uniform_boxes(gen, count, extent=900.0):
    xy = torch.rand((count, 2), generator=gen) * extent
    wh = 8.0 + torch.rand((count, 2), generator=gen) * 192.0
    return torch.cat([xy, xy + wh], dim=1)


@pytest.mark.parametrize("loss_type", ["giou", "l1", "smooth_l1"])
@pytest.mark.parametrize("n,shared_anchors", [(5000, True), (2048, False), (4099, True)])
def test_box_loss_kernels(device, loss_type, n, shared_anchors):
    """Fused regression loss (forward and backward) against the torch restatement of
    RetinaNetRegressionHead.compute_loss evaluated in fp32 like the reference path (the centre differences of the
    encoder cancel, so an fp64 evaluation differs from any fp32 one by more than the tolerance): loss 1e-5 relative,
    gradient 1e-5 of its largest entry plus 1e-4 relative per element, exact zeros for anchors that are not foreground.  Covers a partial last chunk,
    one anchor table for all images and one per image, an image without foreground, size deltas above the clamp
    (no gradient there) and non-unit coder weights."""
    g = torch.Generator().manual_seed(4000 + n + len(loss_type))
    B, M = 3, 11
    weights = (1.0, 1.0, 1.0, 1.0) if loss_type == "giou" else (10.0, 10.0, 5.0, 5.0)
    anchors = _uniform_boxes(g, n) if shared_anchors else torch.stack([_uniform_boxes(g, n) for _ in range(B)])
    gt = [_uniform_boxes(g, M) for _ in range(B)]
    matched = torch.randint(-2, M, (B, n), generator=g)
    matched[2] = torch.where(matched[2] >= 0, torch.full_like(matched[2], -1), matched[2])
    # anchors matched to a target overlap it in a typical way: target = anchor moved and scaled a little
    for b in range(B - 1):
        a = anchors if shared_anchors else anchors[b]
        pick = torch.nonzero(matched[b] == 0).flatten()[:1]
        if pick.numel():
            gt[b][0] = a[pick[0]] + torch.tensor([3.0, -2.0, 9.0, 4.0])
    reg = torch.randn((B, n, 4), generator=g) * 0.5
    reg[0, ::97, 2] = 5.0                                        # above log(1000/16): clamped, zero gradient
    reg[1, 5::131, 3] = 4.5
    coder = tv.BoxCoder(weights)
    x = reg.clone().requires_grad_(True)
    targets = [{"boxes": b} for b in gt]
    a_list = [(anchors if shared_anchors else anchors[b]) for b in range(B)]
    ref = tv.retinanet_head_loss_regression(targets, x, a_list, list(matched), loss_type, 0.3, coder,
                                            eps=float(torch.finfo(torch.float32).eps))
    ref.backward()
    code = {"l1": _C.BOX_LOSS_L1, "smooth_l1": _C.BOX_LOSS_SMOOTH_L1, "giou": _C.BOX_LOSS_GIOU}[loss_type]
    conf = (code, 0.3, weights, coder.bbox_xform_clip, float(torch.finfo(torch.float32).eps))
    rd, ad, md = reg.to(device), anchors.to(device), matched.to(device)
    gd, od = torch.cat(gt).to(device), kernels._device_ints([0, M, 2 * M, 3 * M], device)
    loss, per_image, nfg = kernels.box_loss_forward(rd, ad, md, gd, od, *conf)
    assert nfg.cpu().tolist() == [int((matched[b] >= 0).sum()) for b in range(B)]
    assert per_image[2].item() == 0.0
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item()), (loss.item(), ref.item())
    grad = kernels.box_loss_backward(rd, ad, md, gd, od, *conf, nfg, torch.full((1,), 2.0, device=device)).cpu()
    gr = 2.0 * x.grad
    tol = 1e-5 * gr.abs().max() + 1e-4 * gr.abs()
    assert bool(((grad - gr).abs() <= tol).all()), float((grad - gr).abs().max())
    assert torch.equal(grad[matched < 0], torch.zeros_like(grad[matched < 0]))
    if loss_type == "giou":
        assert float(grad[0, ::97, 2].abs().max()) == 0.0


@pytest.mark.parametrize("num_anchors,num_classes,with_ctr,large", [(9, 4, False, False), (3, 91, False, False),
                                                                    (1, 5, True, False), (2, 3, True, False),
                                                                    (9, 4, False, True), (2, 3, True, True)])
def test_select_candidates_reads_channel_major_head_outputs(device, num_anchors, num_classes, with_ctr, large):
    """od_select_candidates_planar on the convolution layout [B, A*K, H, W] returns exactly what
    od_select_candidates returns on the permuted [B, H*W*A, K] copies (boxes, scores, labels, flat indices, counts),
    for square and ragged planes, chunks that end inside a plane and planes smaller than a chunk."""
    g = torch.Generator().manual_seed(77 + num_anchors + num_classes)
    B, K = 3, 200
    # `large`: a level big enough for the sampled score bound, whose sample positions go through the same
    # position -> flat index conversion
    planes = [(200, 168), (6, 5)] if large else [(48, 40), (17, 23), (6, 5), (1, 1)]
    A, C = num_anchors, num_classes

    def permuted(m, per_anchor):
        b, _, h, w = m.shape
        return m.view(b, -1, per_anchor, h, w).permute(0, 3, 4, 1, 2).reshape(b, -1, per_anchor)

    cls_maps, reg_maps, ctr_maps, anchors = [], [], [], []
    for h, w in planes:
        cls_maps.append((torch.randn((B, A * C, h, w), generator=g) * 1.5 - 1.0).to(device))
        reg = torch.randn((B, A * 4, h, w), generator=g) * 0.5
        reg_maps.append((reg.abs() if with_ctr else reg).to(device))
        ctr_maps.append(torch.randn((B, A, h, w), generator=g).to(device))
        anchors.append(_uniform_boxes(g, h * w * A, extent=400.0).to(device))
    hw = torch.tensor([[500, 480], [400, 500], [512, 512]], dtype=torch.int32, device=device)
    coder = _C.CODER_LINEAR if with_ctr else _C.CODER_BOXCODER
    want = kernels.select_candidates([permuted(m, C) for m in cls_maps], [permuted(m, 4) for m in reg_maps], anchors, hw,
                                     0.3, K, coder, 4.135, want_index=True,
                                     ctr_levels=[permuted(m, 1) for m in ctr_maps] if with_ctr else None)
    got = kernels.select_candidates_planar(cls_maps, reg_maps, anchors, hw, C, 0.3, K, coder, 4.135, want_index=True,
                                           ctr_maps=ctr_maps if with_ctr else None)
    counts = want[4].cpu()
    assert torch.equal(got[4].cpu(), counts) and int(counts.sum()) > 0
    # the channel-major route against the scalar checker run on the permuted copies (not only against the other route)
    g_scores, g_labels, g_index = got[1].cpu(), got[2].cpu(), got[3].cpu()
    for lvl, (cm, tm) in enumerate(zip(cls_maps, ctr_maps)):
        logits = permuted(cm, C).cpu()
        ctr = permuted(tm, 1).cpu() if with_ctr else None
        for b in range(B):
            s_ref, i_ref = c_oracle.filter_topk(logits[b].numpy(), ctr[b].numpy().reshape(-1) if with_ctr else None, C, 0.3, K)
            n = int(counts[b, lvl])
            assert n == len(s_ref), (lvl, b)
            sl = slice(lvl * K, lvl * K + n)
            assert np.array_equal(g_index[b, sl].numpy(), i_ref), (lvl, b)
            assert np.array_equal(g_scores[b, sl].numpy().view(np.uint32), s_ref.view(np.uint32)), (lvl, b)
            assert np.array_equal(g_labels[b, sl].numpy(), i_ref % C), (lvl, b)
    valid = (torch.arange(K)[None, None, :] < counts[:, :, None]).reshape(B, -1)
    for a, b, name in zip(got[:4], want[:4], ("boxes", "scores", "labels", "index")):
        assert torch.equal(a.cpu()[valid], b.cpu()[valid]), name


@pytest.mark.parametrize("min_size,max_size,shapes", [
    (160, 160, [(160, 160), (160, 160)]),                         # already at size: the resize is the identity
    (192, 256, [(120, 150), (300, 200), (97, 131)]),              # up- and down-scaling, ragged, padding
    (64, 100, [(401, 37), (64, 64)]),                             # max_size caps the scale
])
def test_preprocess_images_matches_torch_transform(device, min_size, max_size, shapes):
    """Fused normalize + bilinear resize + padded batch against the checker's restatement of torchvision's
    GeneralizedRCNNTransform.forward on the CPU (normalize -> interpolate(bilinear, align_corners=False,
    recompute_scale_factor=True) -> batch_images, resize_boxes for the targets): 1e-5 absolute + 1e-5 relative on the
    pixels, identical sizes, padding exactly zero, target boxes bit-identical (float and int64 boxes)."""
    from object_detection_amd.transform import GeneralizedRCNNTransform
    g = torch.Generator().manual_seed(min_size + len(shapes))
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    tr = GeneralizedRCNNTransform(min_size, max_size, mean, std)
    images = [torch.rand((3, h, w), generator=g) for h, w in shapes]
    targets = [{"boxes": _uniform_boxes(g, 5, extent=50.0), "labels": torch.ones(5, dtype=torch.int64)} for _ in shapes]
    targets[-1]["boxes"] = targets[-1]["boxes"].round().to(torch.int64)           # as the reference's dataset yields them
    ref_tensors, ref_sizes, ref_t = tv.GeneralizedRCNNTransform(min_size, max_size, mean, std).forward(images, targets)
    got, got_t = tr([im.to(device) for im in images], [{k: v.to(device) for k, v in t.items()} for t in targets])
    assert got.image_sizes == ref_sizes
    assert got.tensors.shape == ref_tensors.shape
    diff = (got.tensors.cpu() - ref_tensors).abs()
    assert bool((diff <= 1e-5 + 1e-5 * ref_tensors.abs()).all()), float(diff.max())
    for i, (h, w) in enumerate(ref_sizes):
        assert float(got.tensors[i, :, h:, :].abs().max() if h < got.tensors.shape[2] else 0.0) == 0.0
        assert float(got.tensors[i, :, :, w:].abs().max() if w < got.tensors.shape[3] else 0.0) == 0.0
    for a, b in zip(got_t, ref_t):
        assert a["boxes"].dtype is torch.float32 and torch.equal(a["boxes"].cpu(), b["boxes"])
    # detections rescaled back: transform.postprocess against the checker's
    dets = [{"boxes": t["boxes"].clone()} for t in got_t]
    back = tr.eval().postprocess(dets, got.image_sizes, [tuple(im.shape[-2:]) for im in images])
    want = tv.GeneralizedRCNNTransform.postprocess([{"boxes": t["boxes"]} for t in ref_t], ref_sizes,
                                                   [tuple(im.shape[-2:]) for im in images])
    for a, b in zip(back, want):
        assert torch.equal(a["boxes"].cpu(), b["boxes"])
    if shapes[0] == (160, 160):
        want0 = (images[0] - torch.tensor(mean)[:, None, None]) / torch.tensor(std)[:, None, None]
        assert torch.equal(got.tensors[0].cpu(), want0)             # identity resize: the normalized pixels themselves


def _sampled_positions(E):
    """Positions the candidate selection samples in a level of E logits (E >= 262144): the library's rule restated."""
    S = (E // 64) & ~16383
    S = min(max(S, 16384), 131072)
    runs = S // 16
    window = E // runs
    run = np.arange(runs, dtype=np.uint64)
    h = (run * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    start = run * np.uint64(window) + ((h * np.uint64(window - 15)) >> np.uint64(32))
    return (start[:, None] + np.arange(16, dtype=np.uint64)[None, :]).reshape(-1).astype(np.int64), S


def test_select_sampled_bound_large_levels(device):
    """Levels of at least 262144 logits estimate a score bound from a sample before filtering.  The result must be
    the exact top-k whatever the sample says: (1) dense random logits (half of them pass a 0.05 threshold);
    (2) thousands of saturated ties beyond K; (3) a level built so that the sample over-estimates the bound --
    the best logits sit exactly on the sampled positions -- and the filter has to repeat the level without the bound;
    (4) every logit equal; (5) threshold <= 0."""
    gen = torch.Generator().manual_seed(2024)
    N, C = 70001, 4                                           # E = 280004
    dense = torch.empty((2, N, C)).normal_(-3.0, 1.5, generator=gen)
    _check_select_single_level(device, dense, C, 0.05, 1000)
    _check_select_single_level(device, dense, C, 0.5, 300)
    ties = dense.clone()
    ties[torch.rand((2, N, C), generator=gen) < 0.02] = 25.0            # ~5600 scores of exactly 1.0 per image
    _check_select_single_level(device, ties, C, 0.5, 1000)
    # (3): sampled positions hold the only high logits, so the r-th best sample is far above the true K-th best
    E = N * C
    pos, S = _sampled_positions(E)
    assert S == 16384 and pos.max() < E and len(np.unique(pos)) == S
    flat = torch.full((E,), -6.0)
    others = torch.randperm(E, generator=gen)[:6000]
    flat[others] = 0.1 + 0.9 * torch.rand(6000, generator=gen)          # pass the threshold, below every sample
    flat[torch.from_numpy(pos)] = 2.0 + 4.0 * torch.rand(S, generator=gen)
    extra = torch.randperm(E, generator=gen)[:300]
    flat[extra] = 7.0                                                   # a few of the very best are not in the sample
    adversarial = flat.view(1, N, C)
    _check_select_single_level(device, adversarial, C, 0.5, 1000)
    # the library's scratch buffer shows that the bound was active and left fewer than K valid keys (layout: key
    # slots, then the reserved / valid / repeat counters 256 bytes apart, then 16 bytes of bound per (image, level))
    ws = kernels._workspaces[(torch.device(device).index or 0, torch.cuda.current_stream(device).cuda_stream, "select")]
    keys_bytes = (((E + 31) & ~31) * 8 + 255) & ~255
    counters = ws[keys_bytes:keys_bytes + 3 * 256].view(torch.int32).cpu()
    bound = ws[keys_bytes + 3 * 256:keys_bytes + 3 * 256 + 16].cpu()
    assert int(bound.view(torch.int32)[2]) == 1, "sampled bound inactive"
    assert 0 < int(counters[64]) < 1000 and int(counters[128]) >= S, counters[[0, 64, 128]].tolist()
    _check_select_single_level(device, adversarial, C, 0.5, 4096)
    _check_select_single_level(device, torch.full((1, N, C), 0.25), C, 0.5, 1000)
    _check_select_single_level(device, dense[:1], C, 0.0, 500)
    _check_select_single_level(device, dense[:1], C, -1.0, 500)
    # FCOS scores (sqrt of class and centre-ness sigmoids) are not a function of the class logit alone: the sample is
    # ranked by score
    Nf, Cf = 140003, 2
    fl = torch.empty((2, Nf, Cf)).normal_(-1.0, 1.5, generator=gen)
    fc = torch.empty((2, Nf, 1)).normal_(0.0, 1.0, generator=gen)
    _check_select_single_level(device, fl, Cf, 0.2, 1000, ctrness=fc)
    _check_select_single_level(device, fl, Cf, 0.6, 300, ctrness=fc)


@pytest.mark.parametrize("batch,max_det", [(1, 1), (3, 300), (16, 500), (2, 257)])
def test_pack_detections_is_the_gather_message(device, batch, max_det):
    """od_pack_detections writes, bit for bit, the message the host-side packing builds (rows x1,y1,x2,y2,score,label,
    then the count), and the one-call gather route returns it unchanged when no process group exists."""
    from object_detection_amd import sharding
    gen = torch.Generator().manual_seed(7 * batch + max_det)
    boxes = torch.rand((batch, max_det, 4), generator=gen) * 1024
    scores = torch.rand((batch, max_det), generator=gen)
    labels = torch.randint(0, 1 << 24, (batch, max_det), generator=gen)
    counts = torch.randint(0, max_det + 1, (batch,), generator=gen, dtype=torch.int32)
    ref = torch.cat((sharding.pack_detections(boxes, scores, labels).reshape(batch, -1),
                     counts.to(torch.float32).reshape(batch, 1)), dim=1)
    dev = [t.to(device) for t in (boxes, scores, labels, counts)]
    msg = kernels.pack_detections(*dev)
    assert msg.shape == (batch, 6 * max_det + 1) and torch.equal(msg.cpu(), ref)
    for reuse in (False, True, True):
        packed, cnt = sharding.gather_padded_detections(*dev, reuse_buffers=reuse)
        assert torch.equal(packed.cpu(), ref[:, :-1].reshape(batch, max_det, 6))
        assert torch.equal(cnt.cpu().to(torch.int32), counts)
    dets = sharding.unpack_detections(packed, cnt)
    assert all(torch.equal(d["labels"].cpu(), labels[i, :int(counts[i])]) for i, d in enumerate(dets))
    with pytest.raises(TypeError):
        kernels.pack_detections(dev[0], dev[1], dev[2].to(torch.int32), dev[3])
    with pytest.raises(ValueError):
        kernels.pack_detections(dev[0], dev[1][:, :-1] if max_det > 1 else dev[1][:0], dev[2], dev[3])


def test_select_very_large_level_deferred_reservation(device):
    """Levels of 8 Mi logits or more are filtered by workgroups that walk eight chunks and reserve key slots once:
    (1) sampled bound active, a key or two per chunk; (2) threshold 0: every logit is a candidate (per-wave passes);
    (3) the sample over-estimates the bound and the level is repeated without it; ragged level size, two images."""
    gen = torch.Generator().manual_seed(88)
    N, C = 2200003, 4                                         # E = 8 800 012: not a multiple of the chunk
    E = N * C
    dense = torch.empty((2, N, C)).normal_(-3.0, 1.5, generator=gen)
    dense[1, :, 0] += 1.0
    _check_select_single_level(device, dense, C, 0.05, 1000)
    _check_select_single_level(device, dense[:1], C, 0.0, 500)
    pos, S = _sampled_positions(E)
    assert S == 131072 and pos.max() < E
    flat = torch.full((E,), -6.0)
    flat[torch.randperm(E, generator=gen)[:6000]] = 0.1 + 0.9 * torch.rand(6000, generator=gen)
    # (samples within one logit unit: the bound sits 1e-4 below the located score, which must not admit 1000 samples)
    flat[torch.from_numpy(pos)] = 2.0 + 1.0 * torch.rand(S, generator=gen)
    flat[torch.randperm(E, generator=gen)[:300]] = 7.0
    _check_select_single_level(device, flat.view(1, N, C), C, 0.5, 1000)
    ws = kernels._workspaces[(torch.device(device).index or 0, torch.cuda.current_stream(device).cuda_stream, "select")]
    keys_bytes = (((E + 31) & ~31) * 8 + 255) & ~255
    counters = ws[keys_bytes:keys_bytes + 3 * 256].view(torch.int32).cpu()
    assert 0 < int(counters[64]) < 1000 and int(counters[128]) >= S, counters[[0, 64, 128]].tolist()


def _predict_tail_checkers(boxes, scores, labels, image_wh, model_wh, thr, float_rule):
    """Both restatements of reference predict.py:66-76 for one image; returns the torch one after checking that the
    scalar C route (fp32 products in numpy, orc_nms, truncation) gives the same answer."""
    pred = {"boxes": boxes.clone(), "scores": scores.clone(), "labels": labels.clone()}
    tb, ts, tl, tk = tv.predict_tail(pred, image_wh, model_wh, thr, float_rule)
    ratio = np.array([image_wh[0] / model_wh[0], image_wh[1] / model_wh[1]] * 2, dtype=np.float32)
    scaled = boxes.numpy() * ratio
    ck = c_oracle.nms(scaled, scores.numpy(), thr, float_rule)
    assert np.array_equal(ck, tk.numpy())
    assert np.array_equal(np.trunc(scaled[ck]).astype(np.int32), tb.numpy())
    assert np.array_equal(scaled, pred["boxes"].numpy())          # mul_ left the rescaled boxes in the dictionary
    return tb, ts, tl, tk, pred["boxes"]


@pytest.mark.parametrize("batch,max_det,extent,sorted_scores", [
    (16, 500, 700.0, True),      # the benchmark's detections_per_img
    (3, 300, 150.0, True),       # the reference's default (configs/config_all.py:12), dense: many suppressions
    (4, 1024, 900.0, False),     # the limit, scores in arbitrary order with ties
    (64, 400, 250.0, False),     # many images that several workgroups share (above 128 detections), fewer detections
                                 # than rows, arbitrary order: the words handed from workgroup to workgroup
    (2, 65, 60.0, False),
    (1, 1, 10.0, True),
])
def test_predict_tail_matches_reference_loop(device, batch, max_det, extent, sorted_scores):
    gen = torch.Generator().manual_seed(batch * 1000 + max_det)
    model_wh = (1024, 1024)
    counts = torch.randint(max_det // 2, max_det + 1, (batch,), generator=gen, dtype=torch.int32)
    counts[0] = max_det
    if batch > 1:
        counts[batch - 1] = 0
    boxes = torch.stack([_random_boxes(max_det, gen, extent) for _ in range(batch)])
    scores = torch.rand((batch, max_det), generator=gen)
    scores[:, ::7] = 0.75
    if sorted_scores:
        scores = scores.sort(dim=1, descending=True, stable=True)[0]
    labels = torch.randint(1, 4, (batch, max_det), generator=gen)
    sizes = [(int(640 + 37 * i), int(480 + 101 * i)) for i in range(batch)]          # (w, h) per image
    ratio = torch.tensor([[w / model_wh[0], h / model_wh[1]] * 2 for w, h in sizes], dtype=torch.float32)
    for thr, mode, float_rule in ((0.3, _C.NMS_THRESHOLD_FLOAT, True), (0.5, 0, False)):
        dboxes = boxes.to(device)
        keep, ob, os_, ol, oc = kernels.predict_tail(dboxes, scores.to(device), labels.to(device), counts.to(device),
                                                     ratio.to(device), thr, mode, scale_in_place=True)
        assert ob.dtype == torch.int32
        for i in range(batch):
            c = int(counts[i])
            tb, ts, tl, tk, scaled = _predict_tail_checkers(boxes[i, :c], scores[i, :c], labels[i, :c], sizes[i],
                                                            model_wh, thr, float_rule)
            k = int(oc[i].item())
            assert k == tk.numel(), (i, thr, k, tk.numel())
            assert torch.equal(keep[i, :k].cpu(), tk)
            assert torch.equal(ob[i, :k].cpu(), tb)
            assert torch.equal(os_[i, :k].cpu(), ts)
            assert torch.equal(ol[i, :k].cpu(), tl)
            assert torch.equal(dboxes[i, :c].cpu(), scaled)                      # rescaled in place, bit for bit
            assert torch.equal(dboxes[i, c:].cpu(), boxes[i, c:])                # rows past the count untouched
            assert not ob[i, k:].any() and not os_[i, k:].any() and not ol[i, k:].any()


def test_predict_tail_known_answers_and_limits(device):
    # IoU == float(0.3) exactly: [0,0,10,10] vs [0,0,10,3]: inter 30, union 100 -> fp32 quotient == float(0.3), which is
    # above the double 0.3 (CPU rule suppresses) and not above float(0.3) (CUDA rule keeps); the ratio 0.5 keeps IoU
    b = torch.tensor([[[0., 0., 20., 20.], [0., 0., 20., 6.], [40., 40., 61., 61.]]], device=device)
    s = torch.tensor([[0.9, 0.8, 0.7]], device=device)
    l = torch.tensor([[1, 2, 3]], device=device)
    c = torch.tensor([3], dtype=torch.int32, device=device)
    r = torch.full((1, 4), 0.5, device=device)
    keep, ob, _, ol, oc = kernels.predict_tail(b.clone(), s, l, c, r, 0.3, 0)
    assert int(oc.item()) == 2 and keep[0, :2].tolist() == [0, 2] and ol[0, :2].tolist() == [1, 3]
    assert ob[0, :2].tolist() == [[0, 0, 10, 10], [20, 20, 30, 30]]             # 30.5 truncates to 30
    keep, _, _, _, oc = kernels.predict_tail(b.clone(), s, l, c, r, 0.3, _C.NMS_THRESHOLD_FLOAT)
    assert int(oc.item()) == 3 and keep[0, :3].tolist() == [0, 1, 2]
    # negative coordinates truncate towards zero; a degenerate pair (0/0) never suppresses
    b2 = torch.tensor([[[-3.7, -0.2, 5.9, 7.99], [1., 1., 1., 1.], [1., 1., 1., 1.]]], device=device)
    keep, ob, _, _, oc = kernels.predict_tail(b2, s, l, c, torch.ones((1, 4), device=device), 0.3, 0,
                                              scale_in_place=False)
    assert int(oc.item()) == 3 and ob[0, 0].tolist() == [-3, 0, 5, 7]
    with pytest.raises(_C.OdBoxesError):
        n = _C.OD_MAX_PREDICT_TAIL + 1
        kernels.predict_tail(torch.zeros((1, n, 4), device=device), torch.zeros((1, n), device=device),
                             torch.zeros((1, n), dtype=torch.int64, device=device), c[:1], r, 0.3)


def test_predict_tail_workspace_contract(device):
    """od_predict_tail through the raw C ABI: the workspace size grows with the detections per image once an image is
    shared by several workgroups, a workspace that is too small or missing is refused (OD_ERR_WORKSPACE / OD_ERR_ARG)
    and nothing is launched, sizes beyond the limit give 0 / OD_ERR_LIMIT."""
    import ctypes
    lib = _C.lib()
    assert lib.od_predict_tail_workspace_bytes(4, 100) < lib.od_predict_tail_workspace_bytes(4, 500)
    assert lib.od_predict_tail_workspace_bytes(4, _C.OD_MAX_PREDICT_TAIL + 1) == 0
    B, D = 2, 500
    f = lambda *shape: torch.zeros(shape, device=device)
    boxes, scores, ratio = f(B, D, 4), f(B, D), torch.ones((B, 4), device=device)
    boxes[..., 2:] = 10.0                               # every box is [0, 0, 10, 10]
    labels = torch.zeros((B, D), dtype=torch.int64, device=device)
    counts = torch.full((B,), D, dtype=torch.int32, device=device)
    ob = torch.zeros((B, D, 4), dtype=torch.int32, device=device)
    os_, ol, oc = f(B, D), labels.clone(), torch.full((B,), -5, dtype=torch.int32, device=device)
    need = lib.od_predict_tail_workspace_bytes(B, D)
    ws = torch.zeros((need,), dtype=torch.uint8, device=device)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    call = lambda wptr, wbytes, d=D: lib.od_predict_tail(p(boxes), p(scores), p(labels), p(counts), p(ratio), B, d, 0.3, 0,
                                                         None, p(ob), p(os_), p(ol), None, p(oc), wptr, wbytes, None)
    assert call(p(ws), need - 1) == -3
    assert call(None, need) == -1
    assert call(p(ws), need, _C.OD_MAX_PREDICT_TAIL + 1) == -2
    torch.cuda.synchronize()
    assert oc.tolist() == [-5, -5]                      # refused calls launched nothing
    assert call(p(ws), need) == 0
    torch.cuda.synchronize()
    assert oc.tolist() == [1, 1]                        # 500 identical boxes: the first one suppresses the rest
