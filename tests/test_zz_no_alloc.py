"""od_batched_nms_noalloc (scratch memory inside the library, no workspace argument, no allocation) returns, bit for
bit, what od_batched_nms returns: called through the C ABI twice in a row, the second time on a smaller input, so
that leftovers of the first call in the shared scratch array would show."""
import ctypes

import pytest
import torch

from object_detection_amd import _C, kernels

pytestmark = pytest.mark.gpu


def _inputs(batch, num_seg, seg_stride, classes, seed, device):
    gen = torch.Generator().manual_seed(seed)
    n = num_seg * seg_stride
    xy = torch.rand((batch, n, 2), generator=gen) * 600.0
    wh = torch.rand((batch, n, 2), generator=gen) * 90.0 + 1.0
    boxes = torch.cat([xy, xy + wh], dim=2)
    scores = torch.rand((batch, n), generator=gen)
    scores[:, ::5] = scores[:, :1]                                   # ties in score
    # pre-sorted segments, as the candidate selection hands them over: (score desc, index asc) inside each segment
    scores = scores.view(batch, num_seg, seg_stride).sort(dim=2, descending=True, stable=True)[0].reshape(batch, n)
    labels = torch.randint(0, classes, (batch, n), generator=gen, dtype=torch.int32)
    counts = torch.randint(seg_stride // 2, seg_stride + 1, (batch, num_seg), generator=gen, dtype=torch.int32)
    counts[0, 0] = 0                                                  # an empty segment
    return [t.to(device).contiguous() for t in (boxes, scores, labels, counts)]


def _outputs(batch, max_keep, device):
    return (torch.full((batch, max_keep), -7, dtype=torch.int64, device=device),
            torch.full((batch, max_keep, 4), -7.0, device=device),
            torch.full((batch, max_keep), -7.0, device=device),
            torch.full((batch, max_keep), -7, dtype=torch.int64, device=device),
            torch.full((batch,), -7, dtype=torch.int32, device=device))


def _run_both(device, batch, num_seg, seg_stride, classes, max_keep, thr, mode, seed, sorted_segments=1):
    lib = _C.lib()
    boxes, scores, labels, counts = _inputs(batch, num_seg, seg_stride, classes, seed, device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    want = _outputs(batch, max_keep, device)
    nbytes = lib.od_batched_nms_workspace_bytes(batch, num_seg, seg_stride)
    assert 0 < nbytes <= _C.OD_NMS_NOALLOC_SCRATCH_BYTES
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    if not sorted_segments:
        perm = torch.randperm(scores.shape[1], generator=torch.Generator().manual_seed(1000 + seed))
        scores = scores[:, perm.to(device)].contiguous()
    rc = lib.od_batched_nms(p(boxes), p(scores), p(labels), p(counts), num_seg, seg_stride, sorted_segments, batch, thr, mode,
                            max_keep,
                            *[p(t) for t in want], p(ws), nbytes, stream)
    assert rc == 0
    got = _outputs(batch, max_keep, device)
    rc = lib.od_batched_nms_noalloc(p(boxes), p(scores), p(labels), p(counts), num_seg, seg_stride, sorted_segments, batch,
                                    thr, mode, max_keep, *[p(t) for t in got], stream)
    assert rc == 0
    torch.cuda.synchronize(device)
    assert int(want[4].min()) >= 0 and int(want[4].max()) > 0
    for name, a, b in zip(("keep_idx", "boxes", "scores", "labels", "count"), got, want):
        assert torch.equal(a, b), f"{name} differs (batch {batch}, {num_seg} x {seg_stride})"
    return want[4].cpu().tolist()


def test_noalloc_nms_matches_workspace_nms_twice(device):
    # The scratch array holds the global order of unsorted candidates and, when more than 1024 boxes may be kept, the
    # kept lists.  The larger input first: 16 images x 6 x 1000 unsorted candidates keeping up to 3000, then 3 images
    # x 4 x 300 in the same scratch; then the pipeline's shape (sorted segments, 500 kept), which needs no scratch.
    first = _run_both(device, 16, 6, 1000, 4, 3000, 0.3, _C.NMS_AUTO_GPU, seed=1, sorted_segments=0)
    second = _run_both(device, 3, 4, 300, 91, 1100, 0.5, _C.NMS_VANILLA, seed=2, sorted_segments=0)
    third = _run_both(device, 16, 6, 1000, 4, 500, 0.3, _C.NMS_AUTO_GPU | _C.NMS_THRESHOLD_FLOAT, seed=3)
    assert max(first) > 1024 and len(second) == 3 and max(third) == 500
    # beyond the built-in scratch array: rejected, nothing launched
    lib = _C.lib()
    null = ctypes.c_void_p(0)
    assert lib.od_batched_nms_noalloc(null, null, null, null, 8, 8192, 1, 64, 0.5, 0, 10, null, null, null, null, null,
                                      null) == -2


def test_noalloc_nms_optional_outputs_and_graph_replay(device):
    """keep_idx and out_labels are optional (NULL), and after one warm-up call on the device (which resolves the
    address of the scratch array -- not a capturable operation) the entry point can be recorded into a HIP graph:
    the replay, on new input values, gives what od_batched_nms gives."""
    lib = _C.lib()
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    null = ctypes.c_void_p(0)
    batch, num_seg, seg_stride, max_keep = 5, 4, 256, 1200          # unsorted and > 1024 kept: the scratch is in use
    boxes, scores, labels, counts = _inputs(batch, num_seg, seg_stride, 3, 7, device)
    perm = torch.randperm(scores.shape[1], generator=torch.Generator().manual_seed(7)).to(device)
    scores = scores[:, perm].contiguous()
    nbytes = lib.od_batched_nms_workspace_bytes(batch, num_seg, seg_stride)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)

    def reference(bx, sc):
        want = _outputs(batch, max_keep, device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        assert lib.od_batched_nms(p(bx), p(sc), p(labels), p(counts), num_seg, seg_stride, 0, batch, 0.5, _C.NMS_VANILLA,
                                  max_keep, *[p(t) for t in want], p(ws), nbytes, stream) == 0
        torch.cuda.synchronize(device)
        return want

    want = reference(boxes, scores)
    got = _outputs(batch, max_keep, device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    assert lib.od_batched_nms_noalloc(p(boxes), p(scores), p(labels), p(counts), num_seg, seg_stride, 0, batch, 0.5,
                                      _C.NMS_VANILLA, max_keep, null, p(got[1]), p(got[2]), null, p(got[4]), stream) == 0
    torch.cuda.synchronize(device)
    assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2]) and torch.equal(got[4], want[4])
    assert bool((got[0] == -7).all()) and bool((got[3] == -7).all())          # the optional outputs were left alone

    # recorded: the call above was the warm-up
    static_boxes, static_scores = boxes.clone(), scores.clone()
    out = _outputs(batch, max_keep, device)
    side = torch.cuda.Stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        rc = lib.od_batched_nms_noalloc(p(static_boxes), p(static_scores), p(labels), p(counts), num_seg, seg_stride, 0,
                                        batch, 0.5, _C.NMS_VANILLA, max_keep, *[p(t) for t in out],
                                        ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream))
    assert rc == 0
    for seed in (8, 9):
        nb, ns, _, _ = _inputs(batch, num_seg, seg_stride, 3, seed, device)
        ns = ns[:, perm].contiguous()
        static_boxes.copy_(nb)
        static_scores.copy_(ns)
        graph.replay()
        torch.cuda.synchronize(device)
        want = reference(nb, ns)
        for name, a, b in zip(("keep_idx", "boxes", "scores", "labels", "count"), out, want):
            assert torch.equal(a, b), f"replay, seed {seed}: {name} differs"
        assert int(want[4].max()) > 0
