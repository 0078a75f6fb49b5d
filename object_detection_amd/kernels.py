"""Tensor-level launchers for the HIP kernels of ``libod_boxes.so``.

PyTorch is used here only for device memory (outputs and workspaces are ``torch.empty`` tensors on
the inputs' device) and for the current HIP stream.  Every function requires tensors that live on
a GPU; none of them has a CPU implementation.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _C

_workspaces: Dict[tuple, Tensor] = {}
_small_cache: Dict[tuple, Tensor] = {}
_ws_bytes: Dict[tuple, int] = {}     # answers of the *_workspace_bytes queries (pure functions of the sizes)
# While a step is being recorded into a HIP graph (graph_step.capture) this is a list that receives every cached
# tensor the launchers hand to a kernel -- uploaded tables, scratch buffers, reusable candidate buffers.  The recorded
# launches hold raw addresses of them, and the caches above drop entries on their own schedule (the table cache is
# cleared when it is full, a scratch buffer is replaced by a larger one): the graph keeps its own references.
_capture_refs: Optional[list] = None


def _held(t):
    if _capture_refs is not None:
        _capture_refs.append(t)
    return t


def _require_gpu(t: Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor: object_detection_amd has no CPU code path "
                           f"(got device {t.device})")


def _raw_stream(device: torch.device) -> int:
    """Handle of the current HIP stream of ``device`` (the integer torch keeps; no Stream object is built)."""
    if device.type != "cuda":
        raise RuntimeError(f"inputs must be GPU tensors: object_detection_amd has no CPU code path (got device {device})")
    return torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())


def _stream(device: torch.device) -> ctypes.c_void_p:
    return ctypes.c_void_p(_raw_stream(device))


def _ptr(t: Optional[Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _workspace(device: torch.device, tag: str, nbytes: int) -> Tensor:
    """Grow-only scratch buffer per (device, stream, tag); torch returns 256-byte aligned blocks.

    Keyed by the current stream as well: calls enqueued on different streams may run concurrently and
    must not share scratch memory."""
    key = (device.index, _raw_stream(device), tag)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return _held(ws)


def _f32c(t: Tensor) -> Tensor:
    if t.dtype != torch.float32:
        t = t.float()          # fp16 anchors / head outputs are upcast exactly, like torchvision's _upcast
    return t.contiguous()


_pinned_sources: list = []     # pinned host tensors of the uploads below, kept as long as the cached device copies


def _upload(host: Tensor, device: torch.device) -> Tensor:
    """Asynchronous host -> device copy of a small table through pinned memory.  The pinned source is kept with the
    cached device copy: the copy is only enqueued here, and a copy recorded into a HIP graph reads its source again at
    every replay (callers that capture graphs fill the cache before the capture starts, see ``graph_step``)."""
    src = host.pin_memory()
    _pinned_sources.append(src)
    return src.to(device, non_blocking=True)


def _device_ints(values: Sequence[int], device: torch.device, dtype=torch.int32) -> Tensor:
    """Small host tuple -> device tensor, cached by value (avoids a host->device copy per call)."""
    key = (device.index, _raw_stream(device), dtype, tuple(int(v) for v in values))     # per stream: the upload is asynchronous
    t = _small_cache.get(key)
    if t is None:
        if len(_small_cache) > 4096:
            _small_cache.clear()
            _pinned_sources.clear()
        # through pinned memory and an asynchronous copy: a copy from pageable memory would make the host wait for
        # everything queued on the stream (per step, when the GT counts change every step)
        t = _upload(torch.tensor(list(key[3]), dtype=dtype), device)
        _small_cache[key] = t
    return _held(t)


def _device_floats(values: Sequence[float], device: torch.device) -> Tensor:
    """Small host tuple of floats -> fp32 tensor on ``device``, cached by value (per stream on a GPU: the upload is
    asynchronous; host tensors -- target boxes that were not moved to the GPU yet -- get a plain tensor)."""
    vals = tuple(float(v) for v in values)
    if device.type != "cuda":
        return torch.tensor(vals, dtype=torch.float32, device=device)
    key = (device.index, _raw_stream(device), "f32", vals)
    t = _small_cache.get(key)
    if t is None:
        if len(_small_cache) > 4096:
            _small_cache.clear()
            _pinned_sources.clear()
        t = _upload(torch.tensor(list(vals), dtype=torch.float32), device)
        _small_cache[key] = t
    return _held(t)


# ------------------------------------------------------------------------------------------------
_grid_tables: Dict[tuple, "ctypes.Array"] = {}     # host tables of anchor grids, kept alive per geometry


def anchor_generate(grid_sizes: Sequence[Tuple[int, int]], strides: Sequence[Tuple[int, int]],
                    cell_anchors: Sequence[Tensor], device: torch.device) -> Tensor:
    """All pyramid anchors, [N,4] fp32 on ``device``, order (level, y, x, a)."""
    L = len(grid_sizes)
    I32 = ctypes.c_int32 * L
    gh = I32(*[int(g[0]) for g in grid_sizes])
    gw = I32(*[int(g[1]) for g in grid_sizes])
    sh = I32(*[int(s[0]) for s in strides])
    sw = I32(*[int(s[1]) for s in strides])
    nb = I32(*[int(c.shape[0]) for c in cell_anchors])
    base = torch.cat([c.detach().to("cpu", torch.float32).reshape(-1, 4) for c in cell_anchors]).contiguous()
    n_total = sum(int(g[0]) * int(g[1]) * int(c.shape[0]) for g, c in zip(grid_sizes, cell_anchors))
    out = torch.empty((n_total, 4), dtype=torch.float32, device=device)
    rc = _C.lib().od_anchor_generate(gh, gw, sh, sw, nb, ctypes.c_void_p(base.data_ptr()), L, _ptr(out), n_total,
                                     _stream(device))
    _C.check(rc, "od_anchor_generate")
    firsts, first = [], 0
    for g, c in zip(grid_sizes, cell_anchors):
        firsts.append((first, int(g[1]) * int(c.shape[0]), int(g[0])))
        first += int(g[0]) * int(g[1]) * int(c.shape[0])
    return set_anchor_grids(out, firsts)


def pack_boxes(box_list: Sequence[Tensor]) -> list:
    """The boxes of a batch in ONE fp32 buffer [sum M, 4], handed back as per-image views in list order.  The matching and
    loss launchers recognise such a list (consecutive views of one buffer) and read the buffer where it is; any other
    list is concatenated at every call.  The input transform hands the target boxes over this way."""
    if len(box_list) == 0:
        return []
    dev = box_list[0].device
    sizes = [int(b.shape[0]) for b in box_list]
    packed = torch.cat([b.reshape(-1, 4).to(torch.float32) for b in box_list]) if sum(sizes) > 0 else \
        torch.zeros((0, 4), dtype=torch.float32, device=dev)
    return list(packed.split(sizes))


def _pack_gt(gt_list: Sequence[Tensor], device: torch.device) -> Tuple[Tensor, Tensor, int]:
    offsets = [0]
    plain = True                      # every tensor already fp32 [M, 4]: concatenate as they are
    for g in gt_list:
        offsets.append(offsets[-1] + g.shape[0])
        plain = plain and g.dtype is torch.float32 and g.dim() == 2
    m_total = offsets[-1]
    if m_total > 0:
        packed = None
        k0 = next(i for i, g in enumerate(gt_list) if g.shape[0] > 0)
        first = gt_list[k0]
        if plain and first.is_cuda:
            # consecutive views of one buffer (pack_boxes): nothing to copy.  Empty views carry no address of
            # their own (data_ptr() is null), so only the views that hold boxes are placed.
            p0 = first.data_ptr()            # the views before it are empty: its offset is 0
            store = first.untyped_storage()
            if all(g.shape[0] == 0 or (g.data_ptr() == p0 + 16 * o and g.is_contiguous() and g.shape[-1] == 4
                                       and g.untyped_storage().data_ptr() == store.data_ptr())
                   for g, o in zip(gt_list, offsets)) and \
                    store.nbytes() - first.storage_offset() * 4 >= 16 * m_total:
                packed = torch.as_strided(first, (m_total, 4), (4, 1), first.storage_offset())
        if packed is not None:
            gt = packed
        elif plain:
            gt = torch.cat(gt_list)   # cat returns a fresh contiguous tensor; empty [0, 4] entries are fine
        else:
            gt = torch.cat([_f32c(g.reshape(-1, 4)) for g in gt_list if g.shape[0] > 0]).contiguous()
    else:
        gt = torch.zeros((1, 4), dtype=torch.float32, device=device)
    return gt, _device_ints(offsets, device), m_total


def set_anchor_grids(anchors: Tensor, grids: Sequence[Tuple[int, int, int]]) -> Tensor:
    """Note on an anchor tensor how its list is ordered: ``(first, row_len, rows)`` per pyramid level (anchors
    [first, first + row_len * rows) are ``rows`` grid rows of ``row_len`` anchors).  ``match_retinanet`` passes the
    note to the matching kernels, which use it to form tighter tiles; results do not depend on it."""
    anchors._od_anchor_grids = tuple((int(f), int(l), int(r)) for f, l, r in grids)
    return anchors


def match_retinanet(anchors: Tensor, gt_list: Sequence[Tensor], high: float, low: float,
                    allow_low_quality: bool = True, grids: Optional[Sequence[Tuple[int, int, int]]] = None) -> Tensor:
    """Fused box_iou + Matcher.  anchors [N,4] (shared) or [B,N,4]; returns int64 [B,N].  ``grids``: see
    ``set_anchor_grids`` (default: the note on the tensor, if any)."""
    _require_gpu(anchors, "anchors")
    dev = anchors.device
    if grids is None:
        grids = getattr(anchors, "_od_anchor_grids", None)
    anchors = _f32c(anchors)
    B = len(gt_list)
    if anchors.dim() == 3:
        n, stride = anchors.shape[1], anchors.shape[1]
    else:
        n, stride = anchors.shape[0], 0
    gt, offsets, m_total = _pack_gt(gt_list, dev)
    _require_gpu(gt, "target boxes")
    matched = torch.empty((B, n), dtype=torch.int64, device=dev)
    lib = _C.lib()
    nbytes = _ws_bytes.get(("match", m_total, B, n))
    if nbytes is None:
        nbytes = _ws_bytes[("match", m_total, B, n)] = lib.od_match_retinanet_workspace_bytes(m_total, B, n)
    ws = _workspace(dev, "match", nbytes)
    if grids:
        key = tuple(grids)
        table = _grid_tables.get(key)
        if table is None:
            table = _grid_tables[key] = (ctypes.c_int64 * (3 * len(key)))(*[int(v) for g in key for v in g])
        rc = lib.od_match_retinanet_grids(_ptr(anchors), stride, n, table, len(key), _ptr(gt), _ptr(offsets), m_total, B,
                                          float(high), float(low), int(bool(allow_low_quality)), _ptr(matched), _ptr(ws),
                                          ws.numel(), _stream(dev))
    else:
        rc = lib.od_match_retinanet(_ptr(anchors), stride, n, _ptr(gt), _ptr(offsets), m_total, B, float(high),
                                    float(low), int(bool(allow_low_quality)), _ptr(matched), _ptr(ws), ws.numel(),
                                    _stream(dev))
    _C.check(rc, "od_match_retinanet")
    return matched


def match_fcos(anchors: Tensor, gt_list: Sequence[Tensor], n_first_level: int, n_last_level: int,
               center_sampling_radius: float) -> Tensor:
    _require_gpu(anchors, "anchors")
    dev = anchors.device
    anchors = _f32c(anchors)
    B = len(gt_list)
    if anchors.dim() == 3:
        n, stride = anchors.shape[1], anchors.shape[1]
    else:
        n, stride = anchors.shape[0], 0
    gt, offsets, _ = _pack_gt(gt_list, dev)
    matched = torch.empty((B, n), dtype=torch.int64, device=dev)
    rc = _C.lib().od_match_fcos(_ptr(anchors), stride, n, int(n_first_level), int(n_last_level), _ptr(gt),
                                _ptr(offsets), B, float(center_sampling_radius), _ptr(matched), _stream(dev))
    _C.check(rc, "od_match_fcos")
    return matched


def box_iou(boxes1: Tensor, boxes2: Tensor) -> Tensor:
    _require_gpu(boxes1, "boxes1")
    _require_gpu(boxes2, "boxes2")
    b1, b2 = _f32c(boxes1), _f32c(boxes2)
    m, n = b1.shape[0], b2.shape[0]
    out = torch.empty((m, n), dtype=torch.float32, device=b1.device)
    for r0 in range(0, max(m, 1), 65535):          # grid.y limit
        r1 = min(m, r0 + 65535)
        if r1 <= r0:
            break
        rc = _C.lib().od_box_iou(_ptr(b1[r0:r1]), r1 - r0, _ptr(b2), n, _ptr(out[r0:r1]), _stream(b1.device))
        _C.check(rc, "od_box_iou")
    return out


def matcher(quality: Tensor, high: float, low: float, allow_low_quality: bool) -> Tensor:
    _require_gpu(quality, "match_quality_matrix")
    q = _f32c(quality)
    m, n = q.shape
    dev = q.device
    out = torch.empty((n,), dtype=torch.int64, device=dev)
    lib = _C.lib()
    ws = _workspace(dev, "matcher", lib.od_matcher_workspace_bytes(m, n))
    rc = lib.od_matcher(_ptr(q), m, n, float(high), float(low), int(bool(allow_low_quality)), _ptr(out), _ptr(ws),
                        ws.numel(), _stream(dev))
    _C.check(rc, "od_matcher")
    return out


def decode_boxes(rel_codes: Tensor, boxes: Tensor, coder: int, bbox_xform_clip: float,
                 clip_hw: Optional[Tuple[int, int]] = None) -> Tensor:
    _require_gpu(rel_codes, "rel_codes")
    rel, bx = _f32c(rel_codes.reshape(-1, 4)), _f32c(boxes.reshape(-1, 4))
    out = torch.empty_like(rel)
    ch, cw = (float(clip_hw[0]), float(clip_hw[1])) if clip_hw is not None else (0.0, 0.0)
    rc = _C.lib().od_decode_boxes(_ptr(rel), _ptr(bx), rel.shape[0], int(coder), float(bbox_xform_clip), ch, cw,
                                  _ptr(out), _stream(rel.device))
    _C.check(rc, "od_decode_boxes")
    return out


class SelectPlan:
    """Host-side argument block of od_select_candidates for one pyramid geometry, built once."""

    def __init__(self, level_sizes: Sequence[int], batch: int, num_classes: int, device: torch.device):
        self.L = len(level_sizes)
        self.level_sizes = [int(s) for s in level_sizes]
        self.batch, self.num_classes, self.device = int(batch), int(num_classes), device
        self.P = ctypes.c_void_p * self.L
        self.I64 = ctypes.c_int64 * self.L
        self.sizes = self.I64(*self.level_sizes)
        self.ws_bytes = _C.lib().od_select_candidates_workspace_bytes(self.sizes, self.L, self.batch, self.num_classes)
        if self.ws_bytes == 0 and self.batch > 0:
            raise _C.OdBoxesError("od_select_candidates_workspace_bytes: shape outside the library limits")


_plans: Dict[tuple, SelectPlan] = {}


_candidate_buffers: Dict[tuple, tuple] = {}


def _candidate_outputs(dev: torch.device, B: int, L: int, K: int, want_index: bool, reuse: bool):
    """Output tensors of the candidate selection.  ``reuse``: per-stream buffers that the next call on the same
    stream overwrites -- for callers that consume the candidates at once on that stream (postprocess: the NMS call
    that follows), so that a step does not allocate five intermediate tensors."""
    key = (dev.index, _raw_stream(dev), B, L, K) if reuse and not want_index else None
    if key is not None:
        bufs = _candidate_buffers.get(key)
        if bufs is not None:
            return _held(bufs)
    bufs = (torch.empty((B, L * K, 4), dtype=torch.float32, device=dev),
            torch.empty((B, L * K), dtype=torch.float32, device=dev),
            torch.empty((B, L * K), dtype=torch.int32, device=dev),
            torch.empty((B, L * K), dtype=torch.int64, device=dev) if want_index else None,
            torch.empty((B, L), dtype=torch.int32, device=dev))
    if key is not None:
        if len(_candidate_buffers) > 64:
            _candidate_buffers.clear()
        _candidate_buffers[key] = bufs
    return _held(bufs)


def select_candidates(cls_levels: Sequence[Tensor], reg_levels: Sequence[Tensor], anchor_levels: Sequence[Tensor],
                      image_hw: Tensor, score_thresh: float, topk: int, coder: int, bbox_xform_clip: float,
                      ctr_levels: Optional[Sequence[Tensor]] = None, want_index: bool = False,
                      reuse_outputs: bool = False):
    """Score filter + top-k + decode + clip for all (image, level).

    cls_levels[l]: [B, N_l, C]; reg_levels[l]: [B, N_l, 4]; anchor_levels[l]: [N_l, 4] (shared by the
    batch) or [B, N_l, 4]; ctr_levels[l]: [B, N_l, 1] for FCOS.  Level tensors may be views of one
    [B, N, .] tensor (inner dims contiguous).  Returns padded candidates, slot = level*topk + rank:
    (boxes [B,L*K,4], scores [B,L*K], labels int32 [B,L*K], index int64 [B,L*K] or None, count int32 [B,L]).
    """
    L = len(cls_levels)
    dev = cls_levels[0].device
    _require_gpu(cls_levels[0], "cls_logits")
    B, C = cls_levels[0].shape[0], cls_levels[0].shape[-1]
    sizes = [int(t.shape[1]) for t in cls_levels]
    key = (dev.index, tuple(sizes), B, C)
    plan = _plans.get(key)
    if plan is None:
        plan = _plans[key] = SelectPlan(sizes, B, C, dev)

    f32 = torch.float32

    def prep(levels, inner):
        out, strides = [], []
        for t in levels:
            if t.dtype is not f32:
                t = t.float()
            if t.dim() == 3:
                st = t.stride()
                if st[2] != 1 or st[1] != inner:
                    t = t.contiguous()
                    st = t.stride()
                strides.append(st[0])
            else:
                strides.append(0)
            out.append(t)
        return out, strides

    cls_t, cls_s = prep(cls_levels, C)
    reg_t, reg_s = prep(reg_levels, 4)
    anc_t, anc_s = prep(anchor_levels, 4)
    anc_t = [a if a.dim() == 3 or a.is_contiguous() else a.contiguous() for a in anc_t]
    if ctr_levels is not None:
        ctr_t, ctr_s = prep(ctr_levels, 1)
    K = int(topk)
    cand_boxes, cand_scores, cand_labels, cand_index, cand_count = _candidate_outputs(dev, B, L, K, want_index,
                                                                                      reuse_outputs)
    ws = _workspace(dev, "select", plan.ws_bytes)
    P, I64 = plan.P, plan.I64
    rc = _C.lib().od_select_candidates(
        P(*[t.data_ptr() for t in cls_t]), I64(*cls_s),
        P(*[t.data_ptr() for t in ctr_t]) if ctr_levels is not None else None,
        I64(*ctr_s) if ctr_levels is not None else None,
        P(*[t.data_ptr() for t in reg_t]), I64(*reg_s),
        P(*[t.data_ptr() for t in anc_t]), I64(*anc_s),
        plan.sizes, L, B, C, _ptr(image_hw), float(score_thresh), K, int(coder), float(bbox_xform_clip),
        _ptr(cand_boxes), _ptr(cand_scores), _ptr(cand_labels), _ptr(cand_index), _ptr(cand_count),
        _ptr(ws), ws.numel(), _stream(dev))
    _C.check(rc, "od_select_candidates")
    return cand_boxes, cand_scores, cand_labels, cand_index, cand_count


def select_candidates_planar(cls_maps: Sequence[Tensor], reg_maps: Sequence[Tensor], anchor_levels: Sequence[Tensor],
                             image_hw: Tensor, num_classes: int, score_thresh: float, topk: int, coder: int,
                             bbox_xform_clip: float, ctr_maps: Optional[Sequence[Tensor]] = None,
                             want_index: bool = False, reuse_outputs: bool = False):
    """select_candidates on the head's convolution outputs as they are: cls_maps[l] [B, A*C, H, W],
    reg_maps[l] [B, A*4, H, W], ctr_maps[l] [B, A, H, W] (FCOS); anchor_levels[l] [H*W*A, 4] in (y, x, a) order.
    Results are those of select_candidates on the permuted [B, H*W*A, .] tensors; the permute copy is not made."""
    L = len(cls_maps)
    dev = cls_maps[0].device
    _require_gpu(cls_maps[0], "cls_logits")
    B, C = cls_maps[0].shape[0], int(num_classes)
    planes = [int(t.shape[2] * t.shape[3]) for t in cls_maps]
    sizes = [int(t.shape[1]) // C * pl for t, pl in zip(cls_maps, planes)]
    key = (dev.index, tuple(sizes), B, C)
    plan = _plans.get(key)
    if plan is None:
        plan = _plans[key] = SelectPlan(sizes, B, C, dev)

    def dense(maps, per_anchor):
        out = []
        for t, n, pl in zip(maps, sizes, planes):
            if t.dtype is not torch.float32:
                t = t.float()
            t = t.contiguous()                               # plain NCHW (a channels_last tensor is copied here)
            if t.shape[0] != B or t.shape[1] * pl != n * per_anchor:
                raise ValueError(f"head output of shape {tuple(t.shape)} does not hold {n} anchors x {per_anchor}")
            out.append(t)
        return out

    cls_t, reg_t = dense(cls_maps, C), dense(reg_maps, 4)
    ctr_t = dense(ctr_maps, 1) if ctr_maps is not None else None
    anc_t, anc_s = [], []
    for a, n in zip(anchor_levels, sizes):
        if a.dtype is not torch.float32:
            a = a.float()
        a = a.contiguous()
        if a.shape[-2] != n:
            raise ValueError(f"{a.shape[-2]} anchors for a level of {n}")
        anc_t.append(a)
        anc_s.append(a.stride(0) if a.dim() == 3 else 0)
    K = int(topk)
    cand_boxes, cand_scores, cand_labels, cand_index, cand_count = _candidate_outputs(dev, B, L, K, want_index,
                                                                                      reuse_outputs)
    ws = _workspace(dev, "select", plan.ws_bytes)
    P, I64 = plan.P, plan.I64
    rc = _C.lib().od_select_candidates_planar(
        P(*[t.data_ptr() for t in cls_t]), I64(*[t.stride(0) for t in cls_t]),
        P(*[t.data_ptr() for t in ctr_t]) if ctr_t is not None else None,
        I64(*[t.stride(0) for t in ctr_t]) if ctr_t is not None else None,
        P(*[t.data_ptr() for t in reg_t]), I64(*[t.stride(0) for t in reg_t]),
        P(*[t.data_ptr() for t in anc_t]), I64(*anc_s),
        plan.sizes, I64(*planes), L, B, C, _ptr(image_hw), float(score_thresh), K, int(coder), float(bbox_xform_clip),
        _ptr(cand_boxes), _ptr(cand_scores), _ptr(cand_labels), _ptr(cand_index), _ptr(cand_count),
        _ptr(ws), ws.numel(), _stream(dev))
    _C.check(rc, "od_select_candidates_planar")
    return cand_boxes, cand_scores, cand_labels, cand_index, cand_count


def batched_nms_padded(boxes: Tensor, scores: Tensor, labels: Optional[Tensor], seg_count: Tensor, seg_stride: int,
                       segments_sorted: bool, iou_threshold: float, mode: int, max_keep: int,
                       want_keep_idx: bool = True, want_labels: bool = True):
    """boxes [B, num_seg*seg_stride, 4], scores [B, .], labels int32 [B, .] or None, seg_count int32 [B, num_seg]."""
    _require_gpu(boxes, "boxes")
    dev = boxes.device
    B = boxes.shape[0]
    num_seg = seg_count.shape[1]
    boxes, scores = _f32c(boxes), _f32c(scores)
    if labels is not None and labels.dtype != torch.int32:
        labels = labels.to(torch.int32)
    if labels is not None:
        labels = labels.contiguous()
    max_keep = int(max_keep)
    keep_idx = torch.empty((B, max_keep), dtype=torch.int64, device=dev) if want_keep_idx else None
    out_boxes = torch.empty((B, max_keep, 4), dtype=torch.float32, device=dev)
    out_scores = torch.empty((B, max_keep), dtype=torch.float32, device=dev)
    out_labels = torch.empty((B, max_keep), dtype=torch.int64, device=dev) if want_labels else None
    out_count = torch.empty((B,), dtype=torch.int32, device=dev)
    lib = _C.lib()
    matrix = bool(int(mode) & _C.NMS_PREFER_MATRIX)
    nbytes = _ws_bytes.get(("nms", B, num_seg, int(seg_stride), matrix))
    if nbytes is None:
        nbytes = lib.od_batched_nms_workspace_bytes(B, num_seg, int(seg_stride))
        if matrix:                                     # 0: the call does not qualify, the flag will be ignored
            nbytes = max(nbytes, lib.od_batched_nms_matrix_workspace_bytes(B, num_seg, int(seg_stride)))
        _ws_bytes[("nms", B, num_seg, int(seg_stride), matrix)] = nbytes
    if nbytes == 0 and B > 0:
        raise _C.OdBoxesError(f"od_batched_nms: {num_seg * seg_stride} boxes per image in {num_seg} segments exceed the "
                              f"limits of {_C.OD_MAX_NMS_BOXES} boxes and 64 segments")
    ws = _workspace(dev, "nms", nbytes)
    rc = lib.od_batched_nms(_ptr(boxes), _ptr(scores), _ptr(labels), _ptr(seg_count.contiguous()), num_seg,
                            int(seg_stride), int(bool(segments_sorted)), B, float(iou_threshold), int(mode), max_keep,
                            _ptr(keep_idx), _ptr(out_boxes), _ptr(out_scores), _ptr(out_labels), _ptr(out_count),
                            _ptr(ws), ws.numel(), _stream(dev))
    _C.check(rc, "od_batched_nms")
    return keep_idx, out_boxes, out_scores, out_labels, out_count


def pack_detections(boxes: Tensor, scores: Tensor, labels: Tensor, counts: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """Padded detections of the local images -> the all-gather message, fp32 [B, 6*D + 1] (one launch).

    boxes [B, D, 4] fp32, scores [B, D] fp32, labels [B, D] int64, counts [B] int32 (the outputs of
    ``batched_nms_padded``).  Row layout: D rows (x1, y1, x2, y2, score, label), then the count."""
    _require_gpu(boxes, "boxes")
    dev = boxes.device
    B, D = int(boxes.shape[0]), int(boxes.shape[1])
    if tuple(boxes.shape) != (B, D, 4) or tuple(scores.shape) != (B, D) or tuple(labels.shape) != (B, D) \
            or tuple(counts.shape) != (B,):
        raise ValueError(f"pack_detections: inconsistent shapes {tuple(boxes.shape)}, {tuple(scores.shape)}, "
                         f"{tuple(labels.shape)}, {tuple(counts.shape)}")
    if labels.dtype != torch.int64 or counts.dtype != torch.int32:
        raise TypeError("pack_detections: labels must be int64 and counts int32")
    for name, t in (("scores", scores), ("labels", labels), ("counts", counts)):
        if t.device != dev:
            raise RuntimeError(f"pack_detections: {name} is on {t.device}, boxes on {dev}")
    boxes, scores = _f32c(boxes), _f32c(scores)
    if out is None:
        out = torch.empty((B, 6 * D + 1), dtype=torch.float32, device=dev)
    elif tuple(out.shape) != (B, 6 * D + 1) or out.dtype != torch.float32 or out.device != dev or not out.is_contiguous():
        raise ValueError("pack_detections: `out` must be a contiguous fp32 [B, 6*D + 1] tensor on the inputs' device")
    rc = _C.lib().od_pack_detections(_ptr(boxes), _ptr(scores), _ptr(labels.contiguous()), _ptr(counts.contiguous()),
                                     B, D, _ptr(out), _stream(dev))
    _C.check(rc, "od_pack_detections")
    return out


def predict_tail(boxes: Tensor, scores: Tensor, labels: Tensor, counts: Tensor, ratio: Tensor, iou_threshold: float,
                 mode: int = _C.NMS_THRESHOLD_FLOAT, scale_in_place: bool = True, want_keep_idx: bool = True):
    """Rescale + class-agnostic NMS + int32 conversion + gathers of the reference's prediction loop
    (``object_detection/predict.py:70-76``) for a padded batch in one launch.

    boxes [B, D, 4] fp32, scores [B, D] fp32, labels [B, D] int64, counts [B] int32 (the padded outputs of
    ``batched_nms_padded``), ratio [B, 4] fp32 on the same device.  With ``scale_in_place`` the valid rows of ``boxes``
    are overwritten with the rescaled boxes, as ``boxes.mul_(ratio)`` does.  Returns (keep_idx [B, D] int64 or None,
    boxes int32 [B, D, 4], scores [B, D], labels [B, D], counts int32 [B]); rows past a count are zero."""
    _require_gpu(boxes, "boxes")
    dev = boxes.device
    B, D = int(boxes.shape[0]), int(boxes.shape[1])
    if tuple(boxes.shape) != (B, D, 4) or tuple(scores.shape) != (B, D) or tuple(labels.shape) != (B, D) \
            or tuple(counts.shape) != (B,) or tuple(ratio.shape) != (B, 4):
        raise ValueError(f"predict_tail: inconsistent shapes {tuple(boxes.shape)}, {tuple(scores.shape)}, "
                         f"{tuple(labels.shape)}, {tuple(counts.shape)}, {tuple(ratio.shape)}")
    if boxes.dtype != torch.float32 or scores.dtype != torch.float32 or labels.dtype != torch.int64 \
            or counts.dtype != torch.int32 or ratio.dtype != torch.float32:
        raise TypeError("predict_tail: boxes / scores / ratio must be fp32, labels int64 and counts int32")
    for name, t in (("boxes", boxes), ("scores", scores), ("labels", labels), ("counts", counts), ("ratio", ratio)):
        if t.device != dev:
            raise RuntimeError(f"predict_tail: {name} is on {t.device}, boxes on {dev}")
        if not t.is_contiguous():
            raise ValueError(f"predict_tail: {name} must be contiguous")
    if D > _C.OD_MAX_PREDICT_TAIL:
        raise _C.OdBoxesError(f"od_predict_tail: {D} detections per image exceed the limit of {_C.OD_MAX_PREDICT_TAIL}")
    keep_idx = torch.empty((B, D), dtype=torch.int64, device=dev) if want_keep_idx else None
    out_boxes = torch.empty((B, D, 4), dtype=torch.int32, device=dev)
    out_scores = torch.empty((B, D), dtype=torch.float32, device=dev)
    out_labels = torch.empty((B, D), dtype=torch.int64, device=dev)
    out_count = torch.empty((B,), dtype=torch.int32, device=dev)
    lib = _C.lib()
    nbytes = _ws_bytes.get(("tail", B, D))
    if nbytes is None:
        nbytes = _ws_bytes[("tail", B, D)] = lib.od_predict_tail_workspace_bytes(B, D)
    ws = _workspace(dev, "tail", nbytes)
    rc = lib.od_predict_tail(_ptr(boxes), _ptr(scores), _ptr(labels), _ptr(counts), _ptr(ratio), B, D,
                             float(iou_threshold), int(mode), _ptr(boxes if scale_in_place else None),
                             _ptr(out_boxes), _ptr(out_scores), _ptr(out_labels), _ptr(keep_idx), _ptr(out_count),
                             _ptr(ws), ws.numel(), _stream(dev))
    _C.check(rc, "od_predict_tail")
    return keep_idx, out_boxes, out_scores, out_labels, out_count


# ------------------------------------------------------------------------------------------------
_DTYPE_CODES = {torch.float32: _C.DTYPE_F32, torch.float16: _C.DTYPE_F16, torch.bfloat16: _C.DTYPE_BF16}


def _dtype_code(t: Tensor, what: str) -> int:
    code = _DTYPE_CODES.get(t.dtype)
    if code is None:
        raise TypeError(f"{what}: element type {t.dtype} is not supported (float32, float16, bfloat16)")
    return code


def focal_loss_forward(cls_logits: Tensor, matched: Tensor, gt_labels: Tensor, gt_offsets: Tensor, alpha: float,
                       gamma: float, normalization: int = _C.LOSS_NORM_PER_IMAGE) -> Tuple[Tensor, Tensor, Tensor]:
    """RetinaNet classification loss for a batch.  cls_logits [B,N,C] fp32 / fp16 / bf16 (inner dims contiguous), matched int64
    [B,N], gt_labels int64 [sum M], gt_offsets int32 [B+1].  Returns (loss [1], per_image [B], num_fg int32 [B])."""
    _require_gpu(cls_logits, "cls_logits")
    dev = cls_logits.device
    B, n, C = cls_logits.shape
    lib = _C.lib()
    ws = _workspace(dev, "focal", lib.od_focal_loss_workspace_bytes(n, C, B))
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    per_image = torch.empty((B,), dtype=torch.float32, device=dev)
    num_fg = torch.empty((B,), dtype=torch.int32, device=dev)
    rc = lib.od_focal_loss_forward(_ptr(cls_logits), _dtype_code(cls_logits, "cls_logits"), cls_logits.stride(0), n, C,
                                   _ptr(matched), _ptr(gt_labels),
                                   _ptr(gt_offsets), B, float(alpha), float(gamma), int(normalization), _ptr(loss),
                                   _ptr(per_image),
                                   _ptr(num_fg), _ptr(ws), ws.numel(), _stream(dev))
    _C.check(rc, "od_focal_loss_forward")
    return loss, per_image, num_fg


def focal_loss_backward(cls_logits: Tensor, matched: Tensor, gt_labels: Tensor, gt_offsets: Tensor, alpha: float,
                        gamma: float, num_fg: Tensor, grad_loss: Tensor,
                        normalization: int = _C.LOSS_NORM_PER_IMAGE) -> Tensor:
    dev = cls_logits.device
    B, n, C = cls_logits.shape
    grad = torch.empty((B, n, C), dtype=cls_logits.dtype, device=dev)
    rc = _C.lib().od_focal_loss_backward(_ptr(cls_logits), _dtype_code(cls_logits, "cls_logits"), cls_logits.stride(0), n, C,
                                         _ptr(matched), _ptr(gt_labels),
                                         _ptr(gt_offsets), B, float(alpha), float(gamma), int(normalization), _ptr(num_fg),
                                         _ptr(grad_loss), _ptr(grad), _stream(dev))
    _C.check(rc, "od_focal_loss_backward")
    return grad


def _box_loss_call_args(bbox_regression: Tensor, anchors: Tensor, matched: Tensor, gt_boxes: Tensor, gt_offsets: Tensor,
                        loss_type: int, beta: float, weights: Sequence[float], clip: float, eps: float):
    B, n, _ = bbox_regression.shape
    anchor_stride = 0 if anchors.dim() == 2 else anchors.stride(0)
    w = (ctypes.c_float * 4)(*[float(x) for x in weights])
    return (_ptr(bbox_regression), _dtype_code(bbox_regression, "bbox_regression"), bbox_regression.stride(0),
            _ptr(anchors), anchor_stride, n, _ptr(matched),
            _ptr(gt_boxes), _ptr(gt_offsets), B, int(loss_type), float(beta), w, float(clip), float(eps))


def box_loss_forward(bbox_regression: Tensor, anchors: Tensor, matched: Tensor, gt_boxes: Tensor, gt_offsets: Tensor,
                     loss_type: int, beta: float, weights: Sequence[float], clip: float,
                     eps: float) -> Tuple[Tensor, Tensor, Tensor]:
    """RetinaNet regression loss for a batch.  bbox_regression [B,N,4] fp32 (inner dims contiguous), anchors [N,4]
    (shared) or [B,N,4], matched int64 [B,N], gt_boxes fp32 [sum M,4], gt_offsets int32 [B+1].
    Returns (loss [1], per_image [B], num_fg int32 [B])."""
    _require_gpu(bbox_regression, "bbox_regression")
    dev = bbox_regression.device
    B, n, _ = bbox_regression.shape
    lib = _C.lib()
    ws = _workspace(dev, "box_loss", lib.od_box_loss_workspace_bytes(n, B))
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    per_image = torch.empty((B,), dtype=torch.float32, device=dev)
    num_fg = torch.empty((B,), dtype=torch.int32, device=dev)
    args = _box_loss_call_args(bbox_regression, anchors, matched, gt_boxes, gt_offsets, loss_type, beta, weights, clip, eps)
    rc = lib.od_box_loss_forward(*args, _ptr(loss), _ptr(per_image), _ptr(num_fg), _ptr(ws), ws.numel(), _stream(dev))
    _C.check(rc, "od_box_loss_forward")
    return loss, per_image, num_fg


def box_loss_backward(bbox_regression: Tensor, anchors: Tensor, matched: Tensor, gt_boxes: Tensor, gt_offsets: Tensor,
                      loss_type: int, beta: float, weights: Sequence[float], clip: float, eps: float, num_fg: Tensor,
                      grad_loss: Tensor) -> Tensor:
    dev = bbox_regression.device
    B, n, _ = bbox_regression.shape
    grad = torch.empty((B, n, 4), dtype=bbox_regression.dtype, device=dev)
    args = _box_loss_call_args(bbox_regression, anchors, matched, gt_boxes, gt_offsets, loss_type, beta, weights, clip, eps)
    rc = _C.lib().od_box_loss_backward(*args, _ptr(num_fg), _ptr(grad_loss), _ptr(grad), _stream(dev))
    _C.check(rc, "od_box_loss_backward")
    return grad


def preprocess_images(images: Sequence[Tensor], resized: Sequence[Tuple[int, int]], mean: Sequence[float],
                      std: Sequence[float], out_hw: Tuple[int, int]) -> Tensor:
    """Normalize + bilinear resize + zero-padded batch in one pass.  images[i]: fp32 [C, h_i, w_i] on the GPU;
    resized[i]: (height, width) of image i after the resize; returns fp32 [B, C, out_hw[0], out_hw[1]]."""
    _require_gpu(images[0], "images")
    dev = images[0].device
    B, C = len(images), int(images[0].shape[0])
    imgs = [im if im.is_contiguous() else im.contiguous() for im in images]
    I32 = ctypes.c_int32 * B
    F = ctypes.c_float * C
    out = torch.empty((B, C, int(out_hw[0]), int(out_hw[1])), dtype=torch.float32, device=dev)
    rc = _C.lib().od_preprocess_images(
        (ctypes.c_void_p * B)(*[im.data_ptr() for im in imgs]), I32(*[int(im.shape[1]) for im in imgs]),
        I32(*[int(im.shape[2]) for im in imgs]), I32(*[int(r[0]) for r in resized]), I32(*[int(r[1]) for r in resized]),
        B, C, F(*[float(m) for m in mean]), F(*[float(v) for v in std]), _ptr(out), int(out_hw[0]), int(out_hw[1]),
        _stream(dev))
    _C.check(rc, "od_preprocess_images")
    return out


def _fcos_loss_call_args(bbox_regression: Tensor, bbox_ctrness: Tensor, anchors: Tensor, matched: Tensor,
                         gt_boxes: Tensor, gt_offsets: Tensor, eps: float):
    B, n, _ = bbox_regression.shape
    anchor_stride = 0 if anchors.dim() == 2 else anchors.stride(0)
    return (_ptr(bbox_regression), _ptr(bbox_ctrness), _dtype_code(bbox_regression, "bbox_regression"),
            bbox_regression.stride(0), bbox_ctrness.stride(0), _ptr(anchors), anchor_stride, n, _ptr(matched),
            _ptr(gt_boxes), _ptr(gt_offsets), B, float(eps))


def fcos_box_loss_forward(bbox_regression: Tensor, bbox_ctrness: Tensor, anchors: Tensor, matched: Tensor,
                          gt_boxes: Tensor, gt_offsets: Tensor, eps: float) -> Tuple[Tensor, Tensor, Tensor]:
    """FCOS regression (GIoU of the linearly decoded boxes) and centre-ness losses for a batch.  bbox_regression
    [B,N,4], bbox_ctrness [B,N] (same element type), anchors [N,4] or [B,N,4] fp32, matched int64 [B,N].
    Returns (loss_regression [1], loss_ctrness [1], num_fg int32 [B] = the batch total, repeated)."""
    _require_gpu(bbox_regression, "bbox_regression")
    dev = bbox_regression.device
    B, n, _ = bbox_regression.shape
    lib = _C.lib()
    ws = _workspace(dev, "box_loss", lib.od_box_loss_workspace_bytes(n, B))
    loss_reg = torch.empty((1,), dtype=torch.float32, device=dev)
    loss_ctr = torch.empty((1,), dtype=torch.float32, device=dev)
    num_fg = torch.empty((B,), dtype=torch.int32, device=dev)
    args = _fcos_loss_call_args(bbox_regression, bbox_ctrness, anchors, matched, gt_boxes, gt_offsets, eps)
    rc = lib.od_fcos_box_loss_forward(*args, _ptr(loss_reg), _ptr(loss_ctr), _ptr(num_fg), _ptr(ws), ws.numel(),
                                      _stream(dev))
    _C.check(rc, "od_fcos_box_loss_forward")
    return loss_reg, loss_ctr, num_fg


def fcos_box_loss_backward(bbox_regression: Tensor, bbox_ctrness: Tensor, anchors: Tensor, matched: Tensor,
                           gt_boxes: Tensor, gt_offsets: Tensor, eps: float, num_fg: Tensor, grad_reg_loss: Tensor,
                           grad_ctr_loss: Tensor) -> Tuple[Tensor, Tensor]:
    dev = bbox_regression.device
    B, n, _ = bbox_regression.shape
    grad_reg = torch.empty((B, n, 4), dtype=bbox_regression.dtype, device=dev)
    grad_ctr = torch.empty((B, n), dtype=bbox_regression.dtype, device=dev)
    args = _fcos_loss_call_args(bbox_regression, bbox_ctrness, anchors, matched, gt_boxes, gt_offsets, eps)
    rc = _C.lib().od_fcos_box_loss_backward(*args, _ptr(num_fg), _ptr(grad_reg_loss), _ptr(grad_ctr_loss), _ptr(grad_reg),
                                            _ptr(grad_ctr), _stream(dev))
    _C.check(rc, "od_fcos_box_loss_backward")
    return grad_reg, grad_ctr
