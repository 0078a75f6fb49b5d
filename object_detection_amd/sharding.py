"""Image-sharded data parallelism for the box pipeline (one process per GPU).

Every step of the pipeline is independent per image, so rank r of R simply takes a contiguous slice
of the batch; training-side matching needs no exchange at all.  Inference ends with ONE collective:
an all-gather of the fixed-size padded detections (``[B_local, D, 6]`` fp32 rows x1,y1,x2,y2,score,label
plus ``[B_local]`` int32 counts), which is RCCL over xGMI when the process group uses the "nccl"
backend.  The reference itself is single-process (``object_detection/train.py:24``); this module is
what replaces "run it on one device" when eight GPUs of a node are available.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def shard_bounds(num_images: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of the images for ``rank`` (first ranks take the remainder)."""
    base, rem = divmod(int(num_images), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_detections(boxes: Tensor, scores: Tensor, labels: Tensor) -> Tensor:
    """[B, D, 4], [B, D], [B, D] -> [B, D, 6] fp32 (labels < 2^24 are exact in fp32)."""
    return torch.cat((boxes, scores.unsqueeze(-1), labels.to(torch.float32).unsqueeze(-1)), dim=-1)


def unpack_detections(packed: Tensor, counts: Tensor) -> List[dict]:
    """Padded rows + counts (int, or the fp32 counts of ``gather_padded_detections``) -> the per-image dictionaries."""
    out = []
    for row, c in zip(packed, (int(v) for v in counts.tolist())):
        out.append({"boxes": row[:c, :4], "scores": row[:c, 4], "labels": row[:c, 5].to(torch.int64)})
    return out


def _exchange(local: Tensor, gathered: Optional[Tensor], group) -> Tensor:
    """All-gather of equally shaped 2-d messages, rank order along dim 0 (``gathered``: optional output buffer)."""
    world = dist.get_world_size(group)
    staged = local.is_cuda and dist.get_backend(group) == "gloo"      # gloo exchanges host memory
    device = local.device
    if staged:
        local, gathered = local.cpu(), None
    if gathered is None:
        gathered = local.new_empty((world * local.shape[0], local.shape[1]))
    try:
        dist.all_gather_into_tensor(gathered, local, group=group)
    except NotImplementedError:                   # a backend without the flat form; any other error (a failed RCCL
        parts = [torch.empty_like(local) for _ in range(world)]    # collective, a shape mismatch) is the caller's to see
        dist.all_gather(parts, local, group=group)
        gathered = torch.cat(parts)
    return gathered.to(device) if staged else gathered


def _distributed(group) -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def all_gather_detections(packed: Tensor, counts: Tensor, group=None) -> Tuple[Tensor, Tensor]:
    """Gather equally-sized local shards ``[B_local, D, 6]`` / ``[B_local]`` from every rank, rank order.

    All ranks must hold the same ``B_local`` (pad the last shard with count 0 rows otherwise).
    """
    if not _distributed(group):
        return packed, counts
    B = packed.shape[0]
    # one message per rank: the detections of an image followed by its count (exact in fp32: counts <= D < 2^24)
    local = torch.cat((packed.reshape(B, -1), counts.to(packed.dtype).reshape(B, 1)), dim=1).contiguous()
    gathered = _exchange(local, None, group)
    g_packed = gathered[:, :-1].reshape((gathered.shape[0],) + tuple(packed.shape[1:]))
    g_counts = gathered[:, -1].to(counts.dtype)
    return g_packed, g_counts


_message_buffers: Dict[tuple, Tuple[Tensor, Optional[Tensor]]] = {}


def gather_padded_detections(boxes: Tensor, scores: Tensor, labels: Tensor, counts: Tensor, group=None,
                             reuse_buffers: bool = False) -> Tuple[Tensor, Tensor]:
    """The inference step's one exchange, from the padded outputs of the NMS: ``boxes [B_local, D, 4]``,
    ``scores [B_local, D]``, ``labels [B_local, D]`` int64, ``counts [B_local]`` int32 ->
    ``(packed [R*B_local, D, 6] fp32, counts [R*B_local] fp32)`` over the R ranks, rank order.

    GPU tensors are packed by one kernel (``od_pack_detections``) and exchanged in one collective.  With
    ``reuse_buffers`` the message and the gathered buffer are kept per stream and overwritten by the next call on the
    same stream (the results are views of them): for loops that consume the detections before the next step."""
    if not boxes.is_cuda:                        # host tensors (the gloo tests): same message, built with torch ops
        g_packed, g_counts = all_gather_detections(pack_detections(boxes, scores, labels), counts, group)
        return g_packed, g_counts.to(torch.float32)
    from . import kernels
    B, D = int(boxes.shape[0]), int(boxes.shape[1])
    world = dist.get_world_size(group) if _distributed(group) else 1
    dev = boxes.device
    key = (dev.index, kernels._raw_stream(dev), B, D, world)
    bufs = _message_buffers.get(key) if reuse_buffers else None
    if bufs is None:
        bufs = (torch.empty((B, 6 * D + 1), dtype=torch.float32, device=dev),
                torch.empty((world * B, 6 * D + 1), dtype=torch.float32, device=dev) if world > 1 else None)
        if reuse_buffers:
            if len(_message_buffers) > 64:
                _message_buffers.clear()
            _message_buffers[key] = bufs
    local = kernels.pack_detections(boxes, scores, labels, counts, out=bufs[0])
    gathered = _exchange(local, bufs[1], group) if world > 1 else local
    return gathered[:, :-1].reshape(gathered.shape[0], D, 6), gathered[:, -1]


class DetectionGatherRing:
    """The same exchange, every ``period`` steps instead of every step: the padded detections of ``period`` consecutive
    steps are packed into the rows of one message buffer (``od_pack_detections`` writes each step's rows in place) and
    travel in ONE all-gather.  A collective call costs the host tens of microseconds and the GPU a latency-bound
    kernel, whatever the message size below a few MB; an evaluation loop only needs every rank's detections when the
    metric is computed (the reference accumulates predictions over the whole validation set,
    ``object_detection/train.py:84-87``), so nothing waits for the individual steps.

    ``add`` returns None until the buffer is full and then what ``flush`` returns: ``(packed [R, steps, B_local, D, 6]
    fp32, counts [R, steps, B_local] fp32)`` over the R ranks, views of a buffer that the next round overwrites.
    Call ``flush`` after the last step.  All calls of one ring must be made on the same stream."""

    def __init__(self, batch: int, detections: int, period: int, device: torch.device, group=None):
        self.B, self.D, self.period, self.group = int(batch), int(detections), max(1, int(period)), group
        self.world = dist.get_world_size(group) if _distributed(group) else 1
        width = 6 * self.D + 1
        self.local = torch.empty((self.period, self.B, width), dtype=torch.float32, device=device)
        self.gathered = (torch.empty((self.world * self.period * self.B, width), dtype=torch.float32, device=device)
                         if self.world > 1 else None)
        self.fill = 0

    def add(self, boxes: Tensor, scores: Tensor, labels: Tensor, counts: Tensor):
        row = self.local[self.fill]
        if boxes.is_cuda:
            from . import kernels
            kernels.pack_detections(boxes, scores, labels, counts, out=row)
        else:
            row[:, :-1] = pack_detections(boxes, scores, labels).reshape(self.B, -1)
            row[:, -1] = counts.to(torch.float32)
        self.fill += 1
        return self.flush() if self.fill == self.period else None

    def flush(self):
        steps, self.fill = self.fill, 0
        if steps == 0:
            return None
        local = self.local[:steps].reshape(steps * self.B, -1)
        if self.world > 1:
            out = self.gathered[: self.world * steps * self.B] if self.gathered is not None else None
            gathered = _exchange(local, out, self.group)
        else:
            gathered = local
        g = gathered.reshape(self.world, steps, self.B, 6 * self.D + 1)
        return g[..., :-1].reshape(self.world, steps, self.B, self.D, 6), g[..., -1]
