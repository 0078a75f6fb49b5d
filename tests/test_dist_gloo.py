"""CPU, 2 processes over gloo: the N>1 path (image sharding + the detections all-gather)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from object_detection_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, d, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gen = torch.Generator().manual_seed(0)                      # same global data on every rank
    boxes = torch.rand((total, d, 4), generator=gen)
    scores = torch.rand((total, d), generator=gen)
    labels = torch.randint(0, 91, (total, d), generator=gen)
    counts = torch.randint(0, d + 1, (total,), generator=gen, dtype=torch.int32)
    lo, hi = sharding.shard_bounds(total, rank, world)
    packed = sharding.pack_detections(boxes[lo:hi], scores[lo:hi], labels[lo:hi])
    g_packed, g_counts = sharding.all_gather_detections(packed, counts[lo:hi])
    full = sharding.pack_detections(boxes, scores, labels)
    ok = torch.equal(g_packed, full) and torch.equal(g_counts, counts)
    dets = sharding.unpack_detections(g_packed, g_counts)
    ok = ok and all(dd["boxes"].shape[0] == int(c) for dd, c in zip(dets, counts))
    ok = ok and all(torch.equal(dd["labels"], labels[i, : int(counts[i])]) for i, dd in enumerate(dets))
    # the one-call route from the padded NMS outputs (host tensors here: same message, built with torch ops)
    p2, c2 = sharding.gather_padded_detections(boxes[lo:hi], scores[lo:hi], labels[lo:hi].to(torch.int64), counts[lo:hi])
    ok = ok and torch.equal(p2, full) and torch.equal(c2.to(torch.int32), counts)
    dets2 = sharding.unpack_detections(p2, c2)
    ok = ok and all(torch.equal(a["boxes"], b["boxes"]) and torch.equal(a["labels"], b["labels"]) for a, b in zip(dets, dets2))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_detections_world2():
    world, total, d = 2, 8, 6
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, total, d, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}
