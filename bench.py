#!/usr/bin/env python3
"""Benchmark of the detection-head box pipeline on MI355X.

One *step* = one pass of the hot path over one synthetic batch that is already resident in HBM:
  inference side: score filter + top-k + decode + clip + class-batched NMS (+ detections all-gather
                  over RCCL when more than one rank runs)
  training side : anchor <-> GT matching for the batch (fused box_iou + Matcher, M GT boxes/image)
through the model API (``postprocess_batched`` = the body of ``RetinaNet.postprocess_detections``, and
``RetinaNet.match_anchors`` = the matching loop of ``compute_loss``).  The detection counts of every step
are copied to pinned host memory asynchronously and read by the host at the end of the timed region; the
rate with a blocking host read after every step is reported beside it (``ms_per_step_blocking_host_read``).
Workload = BASELINE.json configs[1]: RetinaNet, 1024x1024, P2-P7, 9 anchors/location (786,240
anchors/image), 4 classes, topk_candidates=1000, detections_per_img=500, nms_thresh=0.3,
score_thresh=0.5, batch 16.  N > 1: `--scaling weak` (default) gives every rank its own 16 images; `--scaling strong`
cuts ONE global batch of 16 into 16 / N images per rank (sharding.shard_bounds), the split the metric's "bs16 on 1/2/4/8
GPU" describes; the line says which one ran.

Prints ONE JSON line (rank 0): images/s over all ranks, plus ``roofline`` for the dominant kernel
(its duration measured with HIP events on the launch stream during the timed steps) and
``cpu_baseline`` (the CPU restatement of the reference path timed on a bounded sample, N=1 only).
"""
from __future__ import annotations

import argparse
import dataclasses
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from object_detection_amd import _C, kernels, sharding, synthetic  # noqa: E402
from object_detection_amd.retinanet_model import get_model  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak
NUM_CUS = 256            # compute units of one MI355X
# HBM bytes and VALU instructions per launch from the rocprofv3 PMC counters (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU;
# collected by tools/pmc_r03.sh in separate profiler runs of this same script and committed): PMC counters cannot be
# read from inside the process.  The file records the git blob hashes of the kernel sources it was taken from; when
# they differ from the sources next to this script the line says `traffic_stale: true`.
PMC_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03_pmc.json")
KERNEL_SOURCES = ("od_common.h", "od_match.hip", "od_select.hip", "od_nms.hip")


def source_hashes():
    """git blob hash (sha1 of "blob <size>\\0" + content) of every kernel source the counters depend on."""
    import hashlib
    out = {}
    for name in KERNEL_SOURCES:
        try:
            with open(os.path.join(ROOT, "object_detection_amd", "csrc", name), "rb") as f:
                data = f.read()
        except OSError:
            continue
        out[name] = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
    return out


def pmc_table():
    try:
        with open(PMC_FILE) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


# profile id of a launch (od_profile) -> name of the kernel that runs under it for this workload (the one-launch matching
# path reports under the ids of the chain's assign and restore steps)
PMC_KERNEL_NAMES = {"match_assign": ("match_fused_kernel", "match_assign_kernel"),
                    "match_restore": ("match_restore_lds_kernel", "match_restore_kernel")}


def pmc_entry(kernel: str, table=None):
    kernels_ = (table if table is not None else pmc_table()).get("kernels", {})
    for name in PMC_KERNEL_NAMES.get(kernel, (kernel + "_kernel",)):
        if name in kernels_:
            return kernels_[name]
    return None


def pmc_traffic(kernel: str, table=None):
    entry = pmc_entry(kernel, table)
    return None if entry is None else entry.get("hbm_bytes_per_launch")


def minimal_bytes(cfg, kernel: str, n_cand_per_image: float) -> float:
    """Bytes that have to cross HBM at least (per launch of a batch): like algorithmic_bytes, but a table shared by the
    batch (the anchors: anchor_batch_stride = 0) counts once."""
    B, N, M = cfg.batch, cfg.num_anchors, cfg.num_gt
    if kernel == "match_assign":
        return 16.0 * N + B * (16.0 * M + 8.0 * N)
    return algorithmic_bytes(cfg, kernel, n_cand_per_image)


def algorithmic_bytes(cfg, kernel: str, n_cand_per_image: float) -> float:
    """Minimal bytes one launch must move (per batch of cfg.batch images): every input read once, every output written once."""
    B, N, C, M = cfg.batch, cfg.num_anchors, cfg.num_classes, cfg.num_gt
    L, K, D = len(cfg.grids), cfg.topk_candidates, cfg.detections_per_img
    return {
        "filter": 4.0 * B * N * C,                       # every logit read once
        "select": 12.0 * B * L * K + 48.0 * B * L * K,   # candidate keys + (reg, anchor, box out)
        "match_assign": B * (16.0 * N + 16.0 * M + 8.0 * N),   # anchors + GT read, int64 matched index written
        "match_heavy": 0.0,                              # (four-launch chain only) tiles of match_assign done cooperatively: counted there
        "match_init": 4.0 * B * M,                      # (four-launch chain only) start value of every GT row maximum
        "match_restore": 0.0,                            # patches a few anchors; no mandatory traffic
        "sample": 0.0,                                   # reads a 1/64 sample of the large levels; no mandatory traffic
        "filter_redo": 0.0,                              # repeats a level only when its sampled bound was too high
        "nms": B * (28.0 * n_cand_per_image + 8.0 * D),
        "anchor": 16.0 * N,
    }.get(kernel, 0.0)


def build_inputs(cfg, rank: int, dev, images=None):
    """Synthetic batch, generated on the CPU from fixed seeds, then copied to the device.  `images`: the (lo, hi) slice
    of the global batch this rank takes (strong scaling: every rank generates the same global batch and keeps its
    share); default: the whole batch, with other images on every rank."""
    if images is None:
        shifted = dataclasses.replace(cfg, cfg_id=cfg.cfg_id + 100 * rank)   # different images per rank
        head = synthetic.make_head_outputs(shifted)
        targets = synthetic.make_targets(shifted, empty_image=cfg.batch - 1)
    else:
        lo, hi = images
        head = synthetic.make_head_outputs(cfg, images=range(lo, hi))
        targets = synthetic.make_targets(cfg, empty_image=cfg.batch - 1)[lo:hi]
    head = {k: v.to(dev) for k, v in head.items()}
    targets = [{k: v.to(dev) for k, v in t.items()} for t in targets]
    # the target boxes as the model's input transform hands them to compute_loss: per-image views of one buffer
    # (transform.GeneralizedRCNNTransform.forward rescales the boxes of a batch in one buffer)
    for t, view in zip(targets, kernels.pack_boxes([t["boxes"] for t in targets])):
        t["boxes"] = view
    return head, targets


def cpu_baseline(cfg, sample_images: int):
    """Reference CPU path (restated in oracle/tv_reference.py, torch ops, all host threads) on a
    bounded sample of the same workload: `sample_images` images of match + postprocess, taken batch by batch
    (input generation is not timed)."""
    from oracle import tv_reference as tv
    sizes = synthetic.level_sizes(cfg)
    ag = tv.AnchorGenerator(synthetic.retinanet_anchor_sizes(len(cfg.grids)), ((0.5, 1.0, 2.0),) * len(cfg.grids))
    anchors = ag.forward(cfg.image_size, [(g, g) for g in cfg.grids], 1)[0]
    anchors_l = list(anchors.split(sizes))
    t_match = t_post = 0.0
    done = 0
    while done < sample_images:
        count = min(cfg.batch, sample_images - done)
        part = dataclasses.replace(cfg, cfg_id=cfg.cfg_id + 1000 * (done // cfg.batch))      # other images every batch
        head = synthetic.make_head_outputs(part, images=range(count))
        targets = synthetic.make_targets(part)[:count]
        split = {k: list(v.split(sizes, dim=1)) for k, v in head.items()}
        t0 = time.perf_counter()
        tv.retinanet_match([t["boxes"] for t in targets], [anchors] * count, cfg.fg_iou_thresh, cfg.bg_iou_thresh)
        t1 = time.perf_counter()
        tv.retinanet_postprocess(split, [anchors_l] * count, [cfg.image_size] * count, cfg.score_thresh,
                                 cfg.topk_candidates, cfg.nms_thresh, cfg.detections_per_img, math="torch",
                                 device_rule="cpu")
        t2 = time.perf_counter()
        t_match += t1 - t0
        t_post += t2 - t1
        done += count
    return {
        "value": sample_images / (t_match + t_post),
        "unit": "images/s",
        "cores": torch.get_num_threads(),
        "kind": "port",
        "sample": f"{sample_images} images of the same workload (match {t_match:.2f} s + postprocess {t_post:.2f} s), "
                  f"torch CPU ops, {torch.get_num_threads()} threads",
    }


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: this process starts N children of the same script, one rank
    per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, as torch.distributed.run would set
    them), and waits for them.  It runs before anything here has touched the GPU and never touches it itself.
    Rank 0 prints the JSON line on the inherited stdout.  Returns the exit status for the whole job."""
    import socket
    import subprocess
    if not os.environ.get("OD_BENCH_REHEARSAL"):
        have = torch.cuda.device_count()                  # counting devices does not initialise the runtime
        if have < n:
            print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible "
                  f"(OD_BENCH_REHEARSAL=1 rehearses the control flow with every rank on GPU 0)", file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    children = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    status = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            rc = children[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 1
                print(f"bench.py: rank {r} exited with status {rc}; stopping the other ranks", file=sys.stderr)
                for o in pending:
                    children[o].terminate()
        time.sleep(0.05)
    return status


def pmc_valu(kernel: str, table=None):
    """VALU instructions per launch (SQ_INSTS_VALU summed over the waves) from the committed counter pass, or None."""
    entry = pmc_entry(kernel, table)
    return None if entry is None else entry.get("valu_insts_per_launch")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cpu-sample", type=int, default=48,
                    help="images in the CPU-baseline sample (0 = skip); 48 = three batches, 10-15 s of CPU work")
    ap.add_argument("--streams", type=int, default=3, choices=(1, 2, 3, 4),
                    help="eager mode: 2: the inference half and the training half of a step run on HIP streams of their "
                         "own; 3: in addition the inference halves of consecutive steps alternate between two streams")
    ap.add_argument("--rotate", type=int, default=4,
                    help="resident input batches used in rotation by consecutive steps (4 x 403 MB of head outputs: "
                         "several times the 256 MiB last-level cache); rounded up to an even number")
    ap.add_argument("--gather-every", type=int, default=8,
                    help="N > 1: the padded detections of this many consecutive steps of a stream travel in one "
                         "all-gather (1 = an exchange in every step); what is left is exchanged before the clock stops")
    ap.add_argument("--graph", action="store_true",
                    help="timed region: replay the step's recorded HIP graph instead of launching every kernel from the "
                         "host (measured slower on this runtime; without the flag the replay rate is a secondary figure)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=os.environ.get("OD_BENCH_SCALING", "weak"),
                    help="N > 1: weak = every rank processes its own batch of 16 images; strong = one global batch of 16 "
                         "images is cut into 16 / N per rank")
    ap.add_argument("--match-restore-groups", type=int, default=3,
                    help="workgroups per image of the matching path's restore kernel during the overlapped step "
                         "(od_match_restore_groups): each holds a whole compute unit, and beside the inference "
                         "pipeline few of them are better (3: +3 %% over the library's default 16, +7 %% over 8); the "
                         "latency loops at the end of the run use the default")
    ap.add_argument("--lean", action="store_true",
                    help="only warm-up, kernel probe and the timed region (for profiling: the kernel statistics of the "
                         "run are then those of the timed region); the secondary figures are reported as null")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')} "
              f"rank(s)", file=sys.stderr)
        sys.exit(2)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the multi-rank control flow on a machine with one GPU: OD_BENCH_REHEARSAL=1 puts every rank on GPU 0
    # and uses the gloo backend (numbers from such a run mean nothing)
    rehearsal = bool(os.environ.get("OD_BENCH_REHEARSAL"))
    if rehearsal:
        import faulthandler
        faulthandler.dump_traceback_later(170, exit=True)      # a hang ends with the stack of every thread
        local_rank = 0
    import torch.distributed as dist
    # OD_BENCH_DIST_SINGLE=1: a single rank goes through the multi-rank code as well (process group over RCCL, ranks_seen,
    # the ring of all-gathers) -- the way to execute the RCCL calls of an N > 1 run on a machine with one GPU
    dist_on = world > 1 or bool(os.environ.get("OD_BENCH_DIST_SINGLE"))
    if dist_on:
        torch.cuda.set_device(local_rank)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner on stdout when its communicator is created (at the first collective); stdout
        # carries the one JSON line and nothing else, so the banner goes to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                warm = torch.zeros((1,), dtype=torch.int32, device=torch.device("cuda", local_rank))
                dist.all_reduce(warm)
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    use_graph = args.graph

    global_cfg = synthetic.CFG2
    cfg, shard = global_cfg, None
    ranks_seen, backend = 1, None
    if dist_on:
        backend = dist.get_backend()
        ones = torch.ones((1,), dtype=torch.int32, device="cpu" if rehearsal else dev)
        dist.all_reduce(ones)                                   # every rank that joined adds one (over RCCL unless rehearsing)
        ranks_seen = int(ones.item())
        if args.scaling == "strong":
            if global_cfg.batch % world:
                print(f"bench.py: --scaling strong cuts a batch of {global_cfg.batch} images into equal shards; "
                      f"{world} ranks do not divide it", file=sys.stderr)
                sys.exit(2)
            shard = sharding.shard_bounds(global_cfg.batch, rank, world)
            cfg = dataclasses.replace(global_cfg, batch=shard[1] - shard[0])
    L = len(cfg.grids)
    model = get_model(cfg.num_classes, layers=list(range(L - 2)), backbone="pyramid_stub", out_channels=32,
                      score_thresh=cfg.score_thresh, nms_thresh=cfg.nms_thresh,
                      topk_candidates=cfg.topk_candidates, detections_per_img=cfg.detections_per_img,
                      fg_iou_thresh=cfg.fg_iou_thresh, bg_iou_thresh=cfg.bg_iou_thresh,
                      min_size=cfg.image_size[0], max_size=cfg.image_size[0]).to(dev).eval()
    _C.match_restore_groups(args.match_restore_groups)
    sizes = synthetic.level_sizes(cfg)
    grids = [(g, g) for g in cfg.grids]
    strides = [(cfg.image_size[0] // g, cfg.image_size[1] // g) for g in cfg.grids]
    # the tensor AnchorGenerator.forward hands to compute_loss for every image (it carries the generator's note about
    # the order of the anchor list, which the matching kernels use to form their tiles), and its per-level views
    anchors_flat, _ = model.anchor_generator._grid_anchors_flat(grids, strides, dev)
    anchors_l = list(anchors_flat.split(sizes))
    assert anchors_flat.shape[0] == cfg.num_anchors
    # Several resident batches, used in rotation: a step never finds its logits, regression values or GT boxes in the
    # 256 MiB last-level cache because an earlier step left them there (one batch is 403 MB of head outputs).
    n_rot = max(2, args.rotate + (args.rotate & 1))
    batches = []
    for r in range(n_rot):
        head, targets_r = build_inputs(dataclasses.replace(global_cfg, cfg_id=global_cfg.cfg_id + 10 * r), rank, dev, images=shard)
        batches.append(({k: list(v.split(sizes, dim=1)) for k, v in head.items()}, targets_r))
    split, targets = batches[0]
    shapes = [cfg.image_size] * cfg.batch
    anchors_per_image = [anchors_flat] * cfg.batch
    split_anchors = [anchors_l] * cfg.batch

    from object_detection_amd import graph_step
    from object_detection_amd.retinanet import postprocess_batched

    def post_padded(r):
        sp = batches[r][0]
        return postprocess_batched(sp["cls_logits"], sp["bbox_regression"], split_anchors, shapes, model.score_thresh,
                                   model.topk_candidates, model.nms_thresh, model.detections_per_img, _C.CODER_BOXCODER,
                                   model.box_coder.bbox_xform_clip, return_padded=True)

    # ---- eager form of a step: every kernel is launched from the host ------------------------------------------
    # The two halves of a step are independent, so each gets a HIP stream of its own: the narrow kernels of the
    # inference half (top-k: 96 workgroups, NMS: one workgroup per image) then run beside the chip-wide matching
    # kernels instead of leaving most CUs idle.  Scratch buffers are per stream (kernels._workspace).
    # With a third stream the inference halves of consecutive steps alternate between two streams, so that the
    # NMS of one step also runs beside the score filter of the next.
    counts_host = torch.empty((max(args.steps, args.warmup, 3), cfg.batch), dtype=torch.int32).pin_memory()
    count_rows = list(counts_host.unbind(0))              # the row views, made once
    two = args.streams >= 2
    n_post = {1: 1, 2: 1, 3: 2, 4: 2}[args.streams]
    s_posts = [torch.cuda.Stream(dev) for _ in range(n_post)] if two else [torch.cuda.current_stream(dev)]
    s_matches = [torch.cuda.Stream(dev) for _ in range(args.streams - n_post)] if two else [torch.cuda.current_stream(dev)]
    default_stream = torch.cuda.current_stream(dev)
    for st in s_posts + s_matches:
        st.wait_stream(default_stream)

    # N > 1: one ring of message buffers per stream that runs inference halves (sharding.DetectionGatherRing)
    rings = {}

    def exchange(ob, os_, ol, oc):
        st = torch.cuda.current_stream(dev)
        ring = rings.get(st.cuda_stream)
        if ring is None:
            ring = rings[st.cuda_stream] = (st, sharding.DetectionGatherRing(cfg.batch, cfg.detections_per_img,
                                                                             args.gather_every, dev))
        ring[1].add(ob, os_, ol, oc)

    def flush_exchanges():
        for st, ring in rings.values():                    # same order and fill level on every rank
            torch.cuda.set_stream(st)
            ring.flush()

    def eager_step(i=0, mode="padded", single_stream=False):
        """mode "padded": padded outputs, counts copied to pinned memory asynchronously; "api": the model's own
        postprocess_detections (per-image dictionaries, cut to size when first looked at); "api-blocking": the same,
        and the host looks at them right away."""
        r = i % n_rot
        # set_stream, not the `with torch.cuda.stream(...)` context manager: the manager's enter / exit cost the
        # host several microseconds each, twice per step
        torch.cuda.set_stream(default_stream if single_stream or not two else s_posts[i % len(s_posts)])
        if mode == "padded":
            ob, os_, ol, oc = post_padded(r)
            if dist_on:
                exchange(ob, os_, ol, oc)
            count_rows[i % len(count_rows)].copy_(oc, non_blocking=True)
            out = oc
        else:
            out = model.postprocess_detections(batches[r][0], split_anchors, shapes)
            if mode == "api-blocking":
                [(d["boxes"], d["scores"], d["labels"]) for d in out]
        torch.cuda.set_stream(default_stream if single_stream or not two else s_matches[i % len(s_matches)])
        matched = model.match_anchors(batches[r][1], anchors_per_image)
        return matched, out

    # ---- recorded form: one HIP graph per resident batch; inside it the matching half is forked onto a stream of its
    # own.  Graphs of consecutive steps are replayed on two launch streams in turn, so that the narrow tail of one
    # step (top-k, NMS) runs beside the chip-wide kernels of the next.  Every graph was recorded on streams of its own
    # and therefore owns its scratch buffers; a graph is always replayed on the same launch stream.
    launch_streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    graphs_post, graphs_match, accumulators = [], [], []

    def record_graphs():
        # Two graphs per resident batch, the inference half and the matching half: replayed on different streams they
        # overlap like the eager launches do.  (One graph holding both halves as parallel branches was measured
        # slower than the eager issue: its branches did not run side by side.)
        for r in range(n_rot):
            acc = torch.zeros((cfg.batch,), dtype=torch.int32, device=dev)     # detections of all replays, per image

            def recorded_post(r=r, acc=acc):
                ob, os_, ol, oc = post_padded(r)
                acc.add_(oc)
                return ob, os_, ol, oc

            def recorded_match(r=r):
                return model.match_anchors(batches[r][1], anchors_per_image)

            graphs_post.append(graph_step.capture(recorded_post, dev, warmup=2))
            graphs_match.append(graph_step.capture(recorded_match, dev, warmup=2))
            accumulators.append(acc)
        torch.cuda.synchronize(dev)
        for acc in accumulators:
            acc.zero_()

    if use_graph:
        record_graphs()

    def graph_step_run(i):
        r = i % n_rot
        torch.cuda.set_stream(launch_streams[r & 1])
        ob, os_, ol, oc = graphs_post[r].replay()
        if dist_on:
            exchange(ob, os_, ol, oc)
        torch.cuda.set_stream(launch_streams[2])
        matched = graphs_match[r].replay()
        return matched, oc

    step = graph_step_run if use_graph else eager_step

    def sync():
        if dist_on:
            flush_exchanges()                              # detections still waiting in a ring travel now
        torch.cuda.set_stream(default_stream)
        torch.cuda.synchronize(dev)
        if dist_on:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize(dev)

    # Device spin-up (untimed, before the W warm-up steps): the first process on an idle machine was measured to run the
    # same kernels several times slower for its first seconds (clock and power state, first-touch mappings), which
    # says nothing about the code.  Batches of steps are issued until two consecutive batches take the same time
    # within 5 % (at least `at_least` seconds, at most 20 s).
    def spin_up(at_least):
        spin = {"seconds": 0.0, "batches": 0, "first_batch_ms_per_step": None, "last_batch_ms_per_step": None}
        t_spin = time.perf_counter()
        prev = None
        while True:
            sync()
            tb = time.perf_counter()
            for i in range(20):
                step(i)
            sync()
            cur = (time.perf_counter() - tb) / 20
            spin["batches"] += 1
            if spin["first_batch_ms_per_step"] is None:
                spin["first_batch_ms_per_step"] = 1e3 * cur
            spin["last_batch_ms_per_step"] = 1e3 * cur
            waited = time.perf_counter() - t_spin
            stop = ((prev is not None and abs(cur - prev) <= 0.05 * prev and waited >= at_least)
                    or waited >= (2.0 if rehearsal else 20.0))
            if rehearsal:
                print(f"[rank {rank}] spin-up batch {spin['batches']}: {1e3 * cur:.3f} ms/step", file=sys.stderr, flush=True)
            if dist_on:                                 # the steps contain a collective: every rank runs the same number
                import torch.distributed as dist
                flag = torch.tensor([1 if stop else 0], dtype=torch.int32, device="cpu" if rehearsal else dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                stop = bool(flag.item()) or spin["batches"] >= 2000
            if stop:
                break
            prev = cur
        spin["seconds"] = time.perf_counter() - t_spin
        if rehearsal:
            print(f"[rank {rank}] spin-up done: {spin}", file=sys.stderr, flush=True)
        return spin

    spin = spin_up(1.0)

    # which kernel dominates?  (untimed probe with every launch bracketed, eager launches on one stream)
    _C.profile_select(_C.PROFILE_ALL)
    for k in range(len(_C.KERNEL_NAMES)):
        _C.profile_read(k)
    eager_step(single_stream=True)                        # the default stream's scratch buffers
    torch.cuda.synchronize(dev)
    for k in range(len(_C.KERNEL_NAMES)):
        _C.profile_read(k)
    for i in range(3):
        eager_step(i, single_stream=True)                 # one stream: every kernel has the GPU to itself
    torch.cuda.synchronize(dev)
    probe = {}
    for k, name in enumerate(_C.KERNEL_NAMES):
        ms, n = _C.profile_read(k)
        if n:
            probe[name] = {"ms_per_launch": ms / n, "launches_per_step": n / 3}
    _C.profile_select(_C.PROFILE_OFF)
    n_cand = float(kernels_candidates_per_image(cfg, split, split_anchors, shapes, model))
    # Dominant = the kernel that occupies most of the chip per step: duration x the share of the 256 CUs its grid can
    # fill.  NMS (one workgroup per image) and the top-k kernels (one per image and level) are as long as the matching
    # kernel but hold 16 / 96 CUs while the rest of the chip runs other streams' kernels; a chip-level roofline says
    # nothing about them (their own bound is the latency of one workgroup).
    narrow = {"nms": cfg.batch, "select": cfg.batch * L, "sample": cfg.batch * L}
    for name, pr in probe.items():
        pr["cu_share"] = min(1.0, narrow.get(name, NUM_CUS) / NUM_CUS)
    dominant = max(probe, key=lambda name: probe[name]["ms_per_launch"] * probe[name]["launches_per_step"]
                   * probe[name]["cu_share"])
    dom_id = _C.KERNEL_NAMES.index(dominant)
    # The probe leaves the GPU nearly idle for a millisecond; the first tens of steps after such a pause were measured
    # 10 % slower.  A second, short spin-up brings the device back to where the first one left it; the W warm-up
    # steps follow it and run into the timed region without a pause.
    respin = spin_up(0.2)
    spin["seconds"] += respin["seconds"]
    spin["batches"] += respin["batches"]
    spin["last_batch_ms_per_step"] = respin["last_batch_ms_per_step"]

    for i in range(max(args.warmup, 2)):                  # the W warm-up steps of the contract
        step(i)

    # ---- the timed region: exactly K steps ---------------------------------------------------------------------
    # Every 8th launch of the dominant kernel is timed inside it (start / stop events on the dispatch; a timed launch
    # costs the host tens of microseconds, hence not every launch).  Launches inside a recorded graph cannot carry
    # events: with --graph the kernel is timed in a second loop of eager steps after the region.
    if not use_graph:
        _C.profile_select(dom_id)
        _C.profile_stride(8)
        _C.profile_read(dom_id)
    sync()
    if accumulators:                                      # recorded steps count their detections on the device
        for acc in accumulators:
            acc.zero_()
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    host_enqueue = time.perf_counter() - t0               # the host's share: everything issued, GPU possibly still busy
    sync()
    if use_graph:                                         # the host reads the detection counts of all steps here
        total_detections = int(sum(int(acc.sum()) for acc in accumulators))
    else:
        total_detections = int(counts_host[:min(args.steps, len(count_rows))].sum()) * max(1, args.steps // len(count_rows))
    elapsed = time.perf_counter() - t0
    dom_ms, dom_n = (0.0, 0) if use_graph else _C.profile_read(dom_id)
    _C.profile_select(_C.PROFILE_OFF)
    _C.profile_stride(1)
    if rehearsal:
        print(f"[rank {rank}] timed region done", file=sys.stderr, flush=True)

    def timed_loop(fn, n, consume=None, lag=8):
        """ms per call of fn(i), i < n.  `consume(result)` is applied to every result `lag` calls after it was produced
        (a consumer that runs a few steps behind the GPU), to the last ones after the loop."""
        from collections import deque
        fn(0)
        sync()
        pending = deque()
        t1 = time.perf_counter()
        for i in range(n):
            pending.append(fn(i))
            if consume is not None and len(pending) > lag:
                consume(pending.popleft())
        while consume is not None and pending:
            consume(pending.popleft())
        sync()
        return 1e3 * (time.perf_counter() - t1) / n

    # ---- secondary figures (not `value`) ------------------------------------------------------------------------
    ms_timed_kernel = ms_graph = ms_api = ms_api_deferred = ms_api_blocking = ms_post = ms_match = ms_long = None
    if use_graph:
        _C.profile_select(dom_id)
        _C.profile_stride(8)
        _C.profile_read(dom_id)
        ms_timed_kernel = timed_loop(lambda i: eager_step(i), args.steps)
        dom_ms, dom_n = _C.profile_read(dom_id)
        _C.profile_select(_C.PROFILE_OFF)
        _C.profile_stride(1)
    if not args.lean:
        # Through the reference's call signature: model.postprocess_detections(head_outputs, anchors, image_shapes) ->
        # list of per-image dictionaries, which are cut to size when first looked at.  `api`: a consumer looks at
        # the tensors of every image eight steps after the call (by then the counts have arrived: no wait);
        # `api_deferred`: nobody looks inside the loop; `blocking`: the host looks right after every call.
        # the same eager step over 200 steps: a region of K steps costs about (fill + drain of the overlapped streams) +
        # K x the steady-state step, so a short region (the driver times 20 steps) reads higher per step than a long one
        ms_long = timed_loop(lambda i: eager_step(i)[1], 200) if not dist_on else None
        n_api = min(args.steps, 100)
        # (only the detections are held back for the consumer: keeping the 100 MB matched-index tensor of every step
        # alive as well would make the matching half write to fresh memory each step, which is not what is measured)
        look = lambda res: [(d["boxes"], d["scores"], d["labels"]) for d in res]
        ms_api = timed_loop(lambda i: eager_step(i, mode="api")[1], n_api, consume=look)
        ms_api_deferred = timed_loop(lambda i: eager_step(i, mode="api")[1], n_api, consume=lambda res: None, lag=32)
        ms_api_blocking = timed_loop(lambda i: eager_step(i, mode="api-blocking"), 20)
        # the two halves on their own (they never run in the same forward of the reference)
        torch.cuda.set_stream(default_stream)
        ms_post = timed_loop(lambda i: post_padded(i % n_rot), 20)
        ms_match = timed_loop(lambda i: model.match_anchors(batches[i % n_rot][1], anchors_per_image), 20)
        if not use_graph and not dist_on:
            # the recorded form of the step (two HIP graphs per resident batch, replayed on three launch streams)
            record_graphs()
            ms_graph = timed_loop(graph_step_run, min(args.steps, 100))

    # every kernel inside the overlapped step: a loop of eager steps after the timed region with every 8th launch of
    # every kernel timed (start / stop events on the dispatch)
    in_step = {}
    if not args.lean:
        _C.profile_select(_C.PROFILE_ALL)
        _C.profile_stride(8)
        for k in range(len(_C.KERNEL_NAMES)):
            _C.profile_read(k)
        timed_loop(lambda i: eager_step(i), max(64, min(args.steps, 200)))
        for k, name in enumerate(_C.KERNEL_NAMES):
            ms, n = _C.profile_read(k)
            if n:
                in_step[name] = ms / n
        _C.profile_select(_C.PROFILE_OFF)
        _C.profile_stride(1)
    reference_loops = None
    if not args.lean and not dist_on:
        torch.cuda.set_stream(default_stream)
        _C.match_restore_groups(0)                        # latency loops: the library's default
        reference_loops = reference_loop_figures(dev, timed_loop)
        _C.match_restore_groups(args.match_restore_groups)

    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        images = cfg.batch * world * args.steps
        alone_s = probe[dominant]["ms_per_launch"] * 1e-3
        dur_s = (dom_ms / dom_n) * 1e-3 if dom_n else alone_s
        abytes = algorithmic_bytes(cfg, dominant, n_cand)
        achieved = abytes / dur_s / 1e9 if dur_s > 0 else 0.0
        for name, p in probe.items():
            b = algorithmic_bytes(cfg, name, n_cand)
            p["algorithmic_MB"] = b / 1e6
            p["GBps"] = b / (p["ms_per_launch"] * 1e-3) / 1e9 if p["ms_per_launch"] > 0 else 0.0
        is_match = dominant.startswith("match")
        table = pmc_table()
        stale = table.get("source_hashes") != source_hashes()
        traffic = pmc_traffic(dominant, table)
        valu = pmc_valu(dominant, table)
        # every kernel of the step: contract bytes (every input read once and every output written once, per image),
        # minimal bytes (a table shared by the batch counts once), counter traffic, duration alone and inside the
        # overlapped step, and the fraction of the HBM peak each pair gives
        roofline_all = []
        for name, pr in sorted(probe.items(), key=lambda kv: -kv[1]["ms_per_launch"]):
            alone = pr["ms_per_launch"] * 1e-3
            inside = in_step.get(name, 0.0) * 1e-3
            cb, mb, tr = algorithmic_bytes(cfg, name, n_cand), minimal_bytes(cfg, name, n_cand), pmc_traffic(name, table)
            frac = lambda b, t: (b / t / 1e9 / HBM_PEAK_GBS) if (b and t > 0) else None
            roofline_all.append({
                "kernel": name, "launches_per_step": pr["launches_per_step"], "cu_share": pr["cu_share"],
                "us_alone": alone * 1e6, "us_in_step": inside * 1e6 if inside else None,
                "contract_MB": cb / 1e6, "minimal_MB": mb / 1e6, "traffic_MB": tr / 1e6 if tr else None,
                "frac_contract_alone": frac(cb, alone), "frac_contract_in_step": frac(cb, inside),
                "frac_minimal_alone": frac(mb, alone), "frac_minimal_in_step": frac(mb, inside),
                "frac_traffic_alone": frac(tr, alone), "valu_insts": pmc_valu(name, table),
            })
        longest = max(probe, key=lambda name: probe[name]["ms_per_launch"])
        # the whole step against the HBM roofline: contract bytes of all its kernels / time per step; and the matching
        # path as a whole (its contract bytes are moved by all matching kernels together, two launches or four)
        step_bytes = sum(algorithmic_bytes(cfg, name, n_cand) * pr["launches_per_step"] for name, pr in probe.items())
        step_s = elapsed / args.steps
        match_alone_s = sum(pr["ms_per_launch"] * pr["launches_per_step"] for name, pr in probe.items()
                            if name.startswith("match")) * 1e-3
        match_bytes = algorithmic_bytes(cfg, "match_assign", n_cand)
        # a wave64 fp32 instruction occupies its SIMD for 4 cycles; 256 CUs x 4 SIMDs at 2.4 GHz
        valu_issue_s = (valu * 4.0 / (NUM_CUS * 4 * 2.4e9)) if valu else None
        result = {
            "metric": "images/sec (match+decode+NMS) at 1024\u00b2 bs16, 1/2/4/8 GPU; % HBM roofline",
            "value": images / elapsed,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "step_launch": "two recorded HIP graphs per step (inference half, matching half)" if use_graph else "every kernel launched from the host",
            "ms_per_step_host_enqueue": 1e3 * host_enqueue / args.steps,
            "ms_per_step_eager_with_timed_kernel": ms_timed_kernel,
            "ms_per_step_graph_replay": ms_graph,
            # the same work through the reference's call signature (model.postprocess_detections -> per-image
            # dictionaries, model.match_anchors), kernels launched from the host
            "value_api": (cfg.batch * world / (ms_api * 1e-3)) if ms_api else None,
            "ms_per_step_api": ms_api,
            "ms_per_step_api_deferred": ms_api_deferred,
            "ms_per_step_blocking_host_read": ms_api_blocking,
            "detections_per_step": total_detections / args.steps,
            "device_spin_up": spin,
            "phases": None if args.lean else {
                "postprocess_ms": ms_post, "postprocess_images_per_s": cfg.batch / (ms_post * 1e-3),
                "match_ms": ms_match, "match_images_per_s": cfg.batch / (ms_match * 1e-3),
                # brute-force equivalent: every (anchor, GT) pair of the batch counts, although the kernels
                # skip the pairs whose boxes cannot overlap
                "match_pairs_per_s": cfg.batch * cfg.num_anchors * cfg.num_gt / (ms_match * 1e-3),
            },
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak",
            "ranks_seen": ranks_seen,
            "collective_backend": backend,
            "images_per_rank_per_step": cfg.batch,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: RetinaNet 1024x1024 P2-P7 x9 anchors (786240/image), C=4, "
                            "topk_candidates=1000, detections_per_img=500, nms_thresh=0.3, score_thresh=0.5; "
                            f"step = match ({cfg.num_gt} GT/image, fg .6 / bg .5) + postprocess",
                "batch_per_gpu": cfg.batch,
                "global_batch": cfg.batch * world,
                "split": ("one global batch of %d images cut into %d per rank" % (global_cfg.batch, cfg.batch)) if shard else
                         "every rank processes its own batch",
                "anchors_per_image": cfg.num_anchors,
                "candidates_per_image": n_cand,
                "logits": "synthetic 'stress' distribution N(-3, 1.5) + 0.1% saturated: 2.3% of the logits pass "
                          "score_thresh, every level overflows topk_candidates",
                "parallelism": f"image-sharded x{world}" + (
                    (f", all-gather of the padded detections of {args.gather_every} steps at a time (" +
                     ("gloo over host memory: rehearsal on one GPU" if rehearsal else "RCCL") +
                     "): the collective runs once per " + str(args.gather_every) + " steps and stream, so ms_per_step "
                     "holds an eighth of its cost, not a collective per step") if dist_on else ""),
                "resident_batches_in_rotation": n_rot,
                "hip_streams_eager": args.streams,
                "match_restore_groups": args.match_restore_groups,
            },
            "roofline": {
                "kernel": dominant,
                "dominant_by": "duration x share of the 256 CUs the grid can fill (kernels[*].cu_share)",
                "longest_kernel_by_plain_duration": longest,
                "traffic_stale": stale,
                "minimal_MB_per_launch": minimal_bytes(cfg, dominant, n_cand) / 1e6,
                "frac_minimal_bytes": (minimal_bytes(cfg, dominant, n_cand) / dur_s / 1e9 / HBM_PEAK_GBS) if dur_s > 0 else None,
                # the matching kernels are bound by VALU issue, not by HBM: the anchors are shared by the batch and
                # served from L2, so the bytes that cross HBM (`traffic`) are about half the algorithmic bytes
                "bound": "valu" if is_match else "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "frac_moved_bytes": (traffic / dur_s / 1e9 / HBM_PEAK_GBS) if (traffic and dur_s > 0) else None,
                "valu_insts_per_launch": valu,
                "valu_issue_frac": (valu_issue_s / dur_s) if (valu_issue_s and dur_s > 0) else None,
                "launch_us": dur_s * 1e6,
                "launch_us_measured_in": ("a loop of K eager steps after the timed region (launches inside a recorded "
                                          "graph cannot carry events)" if use_graph else
                                          "the timed region: HIP events on every 8th dispatch of this kernel"),
                "algorithmic_MB_per_launch": abytes / 1e6,
                # matching only: 13 fp32 operations per (anchor, GT) pair of the brute-force formulation against the
                # non-FMA VALU peak (78.6 Tops/s); above 1 because the kernels skip the pairs whose boxes cannot overlap
                "valu_frac_bruteforce_equivalent": (13.0 * cfg.batch * cfg.num_anchors * cfg.num_gt / dur_s / 78.6e12
                                                    if is_match and dur_s > 0 else None),
                # the same kernel with the GPU to itself (single-stream probe)
                "launch_us_alone": alone_s * 1e6,
                "frac_alone": (abytes / alone_s / 1e9) / HBM_PEAK_GBS,
                # `launch_us` is the time between the dispatch's start and end stamps while two other streams keep the
                # chip busy: it includes the time the kernel's workgroups wait for compute units that other kernels
                # hold, so it grows with the overlap, not with the kernel's own work (`launch_us_alone`, which is what
                # the kernel statistics under profiles/ show: the tracer slows the host until the streams barely overlap)
                "step_contract_MB": step_bytes / 1e6,
                "step_frac": (step_bytes / step_s / 1e9 / HBM_PEAK_GBS) if step_s > 0 else None,
                "matching_path_us_alone": match_alone_s * 1e6,
                "matching_path_frac_alone": (match_bytes / match_alone_s / 1e9 / HBM_PEAK_GBS) if match_alone_s > 0 else None,
            },
            "kernels": probe,
            "roofline_all": roofline_all,
            "ms_per_step_over_200_steps": ms_long,
            "value_over_200_steps": (cfg.batch / (ms_long * 1e-3)) if ms_long else None,
            "reference_loops": reference_loops,
        }
        if not dist_on and args.cpu_sample > 0:
            result["cpu_baseline"] = cpu_baseline(cfg, args.cpu_sample)
        print(json.dumps(result))

    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def reference_loop_figures(dev, timed_loop):
    """What the reference's own loops see (none of this is `value`): validation and prediction run `model(images)` with
    batch_size 1 and look at the result at once (reference train.py:84-85,104, predict.py:61-76); training runs
    compute_loss forward + backward and reads the loss with .item() every step at batch_size 8
    (configs/config_all.py:3, train.py:62-66).  Thresholds of configs/config_all.py:10-15: topk_candidates 1000,
    detections_per_img 300, score_thresh .5, nms_thresh .3, fg .6 / bg .5; 1024 x 1024, the bench's pyramid and logits."""
    from object_detection_amd.retinanet import postprocess_batched
    out = {}
    ref = dataclasses.replace(synthetic.CFG2, detections_per_img=300)
    L = len(ref.grids)
    sizes = synthetic.level_sizes(ref)
    model = get_model(ref.num_classes, layers=list(range(L - 2)), backbone="pyramid_stub", out_channels=32,
                      score_thresh=0.5, nms_thresh=0.3, topk_candidates=1000, detections_per_img=300,
                      fg_iou_thresh=0.6, bg_iou_thresh=0.5, min_size=1024, max_size=1024).to(dev).eval()
    grids = [(g, g) for g in ref.grids]
    strides = [(ref.image_size[0] // g, ref.image_size[1] // g) for g in ref.grids]
    anchors_flat, _ = model.anchor_generator._grid_anchors_flat(grids, strides, dev)
    anchors_l = list(anchors_flat.split(sizes))
    # ---- validation / prediction: one image per call, every result read at once
    one = dataclasses.replace(ref, batch=1)
    heads = []
    for r in range(4):
        h = synthetic.make_head_outputs(dataclasses.replace(one, cfg_id=one.cfg_id + 50 + r))
        heads.append({k: [t.to(dev) for t in v.split(sizes, dim=1)] for k, v in h.items()})

    def validate(i):
        dets = model.postprocess_detections(heads[i % 4], [anchors_l], [ref.image_size])
        return [(d["boxes"], d["scores"], d["labels"]) for d in dets]       # the first access waits for the GPU

    out["validation_b1_postprocess_ms"] = timed_loop(validate, 200)
    _C.profile_select(_C.PROFILE_ALL)
    for k in range(len(_C.KERNEL_NAMES)):
        _C.profile_read(k)
    for i in range(20):
        validate(i)
    torch.cuda.synchronize(dev)
    chain = {}
    for k, name in enumerate(_C.KERNEL_NAMES):
        ms, n = _C.profile_read(k)
        if n:
            chain[name] = 1e3 * ms / n
    _C.profile_select(_C.PROFILE_OFF)
    out["validation_b1_kernel_us"] = chain
    out["validation_b1_kernel_chain_us"] = sum(chain.values())
    out["validation_b1_images_per_s"] = 1e3 / out["validation_b1_postprocess_ms"]
    # ---- training: compute_loss forward + backward + .item() at the reference's batch size
    tb = dataclasses.replace(ref, batch=8)
    head = {k: v.to(dev).requires_grad_(True) for k, v in synthetic.make_head_outputs(dataclasses.replace(tb, cfg_id=tb.cfg_id + 70)).items()}
    targets = [{k: v.to(dev) for k, v in t.items()} for t in synthetic.make_targets(tb)]
    for t, view in zip(targets, kernels.pack_boxes([t["boxes"] for t in targets])):
        t["boxes"] = view
    model.train()

    def train_step(i):
        for v in head.values():
            v.grad = None
        losses = model.compute_loss(targets, head, [anchors_flat] * tb.batch)
        total = losses["classification"] + losses["bbox_regression"]
        value = total.item()                                               # reference train.py:64
        total.backward()
        return value

    for i in range(30):                                                    # allocator and autograd warm-up (the first
        train_step(i)                                                      # tens of steps were measured 30 % slower)
    out["train_b8_compute_loss_fwd_item_bwd_ms"] = timed_loop(train_step, 100)
    model.eval()
    return out


def kernels_candidates_per_image(cfg, split, split_anchors, shapes, model) -> float:
    """Average number of NMS candidates per image (after score filter + top-k), for byte accounting."""
    from object_detection_amd.retinanet import _image_hw_tensor, _shared_levels
    dev = split["cls_logits"][0].device
    _, _, _, _, cc = kernels.select_candidates(split["cls_logits"], split["bbox_regression"], _shared_levels(split_anchors),
                                               _image_hw_tensor(shapes, dev), model.score_thresh, model.topk_candidates,
                                               _C.CODER_BOXCODER, model.box_coder.bbox_xform_clip)
    return float(cc.sum().item()) / cfg.batch


if __name__ == "__main__":
    main()
