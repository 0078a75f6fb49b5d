"""Image-sharded data parallelism for the box pipeline (one process per GPU).

Every step of the pipeline is independent per image, so rank r of R simply takes a contiguous slice
of the batch; training-side matching needs no exchange at all.  Inference ends with ONE collective:
an all-gather of the fixed-size padded detections (``[B_local, D, 6]`` fp32 rows x1,y1,x2,y2,score,label
plus ``[B_local]`` int32 counts), which is RCCL over xGMI when the process group uses the "nccl"
backend.  The reference itself is single-process (``object_detection/train.py:24``); this module is
what replaces "run it on one device" when eight GPUs of a node are available.
"""
from __future__ import annotations

from typing import List, Tuple

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
    out = []
    for row, c in zip(packed, counts.tolist()):
        out.append({"boxes": row[:c, :4], "scores": row[:c, 4], "labels": row[:c, 5].to(torch.int64)})
    return out


def all_gather_detections(packed: Tensor, counts: Tensor, group=None) -> Tuple[Tensor, Tensor]:
    """Gather equally-sized local shards ``[B_local, D, 6]`` / ``[B_local]`` from every rank, rank order.

    All ranks must hold the same ``B_local`` (pad the last shard with count 0 rows otherwise).
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return packed, counts
    world = dist.get_world_size(group)
    B = packed.shape[0]
    # one message per rank: the detections of an image followed by its count (exact in fp32: counts <= D < 2^24)
    local = torch.cat((packed.reshape(B, -1), counts.to(packed.dtype).reshape(B, 1)), dim=1).contiguous()
    staged = local.is_cuda and dist.get_backend(group) == "gloo"      # gloo exchanges host memory
    device = local.device
    if staged:
        local = local.cpu()
    gathered = local.new_empty((world * B, local.shape[1]))
    try:
        dist.all_gather_into_tensor(gathered, local, group=group)
    except (RuntimeError, NotImplementedError):   # backends without the flat form
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local, group=group)
        gathered = torch.cat(parts)
    if staged:
        gathered = gathered.to(device)
    g_packed = gathered[:, :-1].reshape((world * B,) + tuple(packed.shape[1:]))
    g_counts = gathered[:, -1].to(counts.dtype)
    return g_packed, g_counts
