"""CPU, 2 processes over gloo: the N>1 path (image sharding + the detections all-gather)."""
import os
import socket

import pytest
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
    # the exchange every few steps: 7 steps through a ring of 3 (two full rounds and a flush of one step) give, step by
    # step, what the per-step exchange gives
    steps = 7
    data = []
    for t in range(steps):
        g2 = torch.Generator().manual_seed(100 + t)
        data.append((torch.rand((total, d, 4), generator=g2), torch.rand((total, d), generator=g2),
                     torch.randint(0, 91, (total, d), generator=g2), torch.randint(0, d + 1, (total,), generator=g2, dtype=torch.int32)))
    ring = sharding.DetectionGatherRing(hi - lo, d, 3, torch.device("cpu"))
    seen = []
    for t in range(steps):
        bx, sc, lb, ct = data[t]
        out = ring.add(bx[lo:hi], sc[lo:hi], lb[lo:hi], ct[lo:hi])
        ok = ok and ((out is not None) == (t % 3 == 2))
        if out is not None:
            seen.append((out[0].clone(), out[1].clone()))      # views of the ring's buffer: the next round overwrites them
    out = ring.flush()
    ok = ok and out is not None and ring.flush() is None
    seen.append(out)
    t = 0
    for g_p, g_c in seen:
        ok = ok and g_p.shape[0] == world and g_p.shape[2] == hi - lo
        for k in range(g_p.shape[1]):
            bx, sc, lb, ct = data[t]
            step_packed = g_p[:, k].reshape(total, d, 6)        # rank order = image order (equal shards)
            ok = ok and torch.equal(step_packed, sharding.pack_detections(bx, sc, lb))
            ok = ok and torch.equal(g_c[:, k].reshape(total).to(torch.int32), ct)
            t += 1
    ok = ok and t == steps
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


def _gpu_worker(rank, world, port, ret, backend="gloo"):
    """All ranks on GPU 0 (a one-GPU machine): the image-sharded inference tail -- NMS of the rank's own images,
    od_pack_detections, one all-gather, unpack -- against the NMS of the whole batch.  Two ranks exchange over gloo;
    a single rank can use the real backend ("nccl" = RCCL), which takes the device tensors as they are."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from object_detection_amd import _C, kernels
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(11)                     # the same global batch on every rank
    B, segs, stride, D = 6, 3, 200, 120
    n = segs * stride
    xy = torch.rand((B, n, 2), generator=gen) * 300
    boxes = torch.cat([xy, xy + torch.rand((B, n, 2), generator=gen) * 60 + 1], dim=2)
    scores = torch.rand((B, n), generator=gen).view(B, segs, stride).sort(dim=2, descending=True)[0].reshape(B, n)
    labels = torch.randint(0, 5, (B, n), generator=gen, dtype=torch.int32)
    counts = torch.randint(50, stride + 1, (B, segs), generator=gen, dtype=torch.int32)
    full = [t.to(dev) for t in (boxes, scores, labels, counts)]
    _, fb, fs, fl, fc = kernels.batched_nms_padded(*full, stride, True, 0.4, _C.NMS_AUTO_GPU, D, want_keep_idx=False)
    lo, hi = sharding.shard_bounds(B, rank, world)
    _, ob, os_, ol, oc = kernels.batched_nms_padded(*[t[lo:hi].contiguous() for t in full], stride, True, 0.4,
                                                    _C.NMS_AUTO_GPU, D, want_keep_idx=False)
    ok = True
    for reuse in (False, True, True):
        packed, cnt = sharding.gather_padded_detections(ob, os_, ol, oc, reuse_buffers=reuse)
        ok = ok and packed.is_cuda and tuple(packed.shape) == (B, D, 6)
        ok = ok and torch.equal(cnt.to(torch.int32), fc)
        ok = ok and torch.equal(packed[..., :4], fb) and torch.equal(packed[..., 4], fs)
        ok = ok and torch.equal(packed[..., 5].to(torch.int64), fl)
    dets = sharding.unpack_detections(packed, cnt)
    ok = ok and all(torch.equal(d["boxes"], fb[i, :int(fc[i])]) and torch.equal(d["labels"], fl[i, :int(fc[i])])
                    for i, d in enumerate(dets))
    ok = ok and int(fc.min()) > 0
    # the same through the ring (period 2: one full round, then a flush of one step)
    ring = sharding.DetectionGatherRing(hi - lo, D, 2, dev)
    outs = [ring.add(ob, os_, ol, oc), ring.add(ob, os_, ol, oc)]
    ok = ok and outs[0] is None and outs[1] is not None
    outs[1] = (outs[1][0].clone(), outs[1][1].clone())
    ok = ok and ring.add(ob, os_, ol, oc) is None
    last = ring.flush()
    for g_p, g_c in (outs[1], last):
        for k in range(g_p.shape[1]):
            sp = g_p[:, k].reshape(B, D, 6)
            ok = ok and torch.equal(sp[..., :4], fb) and torch.equal(sp[..., 4], fs) and torch.equal(sp[..., 5].to(torch.int64), fl)
            ok = ok and torch.equal(g_c[:, k].reshape(B).to(torch.int32), fc)
    ok = ok and outs[1][0].shape[1] == 2 and last[0].shape[1] == 1
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_inference_tail_on_gpu_world2():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gpu_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


@pytest.mark.gpu
def test_sharded_inference_tail_over_rccl_single_rank():
    """The same exchange code through the backend a multi-GPU run uses (nccl = RCCL on ROCm), with the one rank a
    one-GPU machine allows: the collective calls, dtypes, buffer reuse and the ring run through RCCL's own entry points
    (a communicator of one rank copies its input to its output)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if not dist.is_nccl_available():
        pytest.skip("this torch build has no nccl / RCCL backend")
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gpu_worker, args=(1, port, ret, "nccl"), nprocs=1, join=True)
    assert dict(ret) == {0: True}


@pytest.mark.gpu
@pytest.mark.parametrize("scaling,per_rank", [("strong", 8), ("weak", 16)])
def test_bench_two_rank_rehearsal(scaling, per_rank):
    """`bench.py --gpus 2` on a one-GPU machine (OD_BENCH_REHEARSAL=1: both ranks on GPU 0, gloo): the launcher, the
    shard of the global batch (strong) or the per-rank batch (weak), the ring of exchanges and the one JSON line of
    rank 0, which says how many ranks joined the collective and how the batch was split."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OD_BENCH_REHEARSAL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--scaling", scaling, "--steps", "16",
                          "--warmup", "2", "--cpu-sample", "0", "--lean"], env=env, capture_output=True, text=True, timeout=280)
    assert run.returncode == 0, run.stderr[-2000:]
    line = json.loads([l for l in run.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["scaling"] == scaling
    assert line["collective_backend"] == "gloo" and line["images_per_rank_per_step"] == per_rank
    assert line["config"]["global_batch"] == 2 * per_rank and line["steps"] == 16
    assert line["value"] > 0 and abs(line["value"] - 2 * per_rank * 16 / (line["ms_per_step"] * 16e-3)) < 1e-6 * line["value"]
    assert line["detections_per_step"] > 0


@pytest.mark.gpu
def test_bench_single_rank_through_rccl():
    """`OD_BENCH_DIST_SINGLE=1 python bench.py`: one rank takes the multi-rank code path over the real backend (process
    group, the all-reduce that counts the ranks, the ring of all-gathers over RCCL).  stdout carries the one JSON line
    and nothing else -- RCCL's version banner included."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if not dist.is_nccl_available():
        pytest.skip("this torch build has no nccl / RCCL backend")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OD_BENCH_DIST_SINGLE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OD_BENCH_REHEARSAL"):
        env.pop(k, None)
    run = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "16", "--warmup", "2", "--cpu-sample", "0",
                          "--lean"], env=env, capture_output=True, text=True, timeout=280)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), run.stdout[:500]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["collective_backend"] == "nccl"
    assert "RCCL" in line["config"]["parallelism"] and line["value"] > 0 and line["detections_per_step"] > 0
