"""HIP-graph capture of a step of the box pipeline (``torch.cuda.CUDAGraph`` is hipGraph on ROCm).

A step of the pipeline is a dozen short kernels; issued one by one the host needs about as long to enqueue them as the
GPU needs to run them.  Captured once and replayed, a step costs the host one launch.

What makes a step safe to capture (and what ``capture`` arranges):
  * every small table the launchers upload (``kernels._device_ints`` / ``_device_floats``: GT offsets, image sizes,
    box ratios) and every scratch buffer (``kernels._workspace``, the reusable candidate buffers) is cached per
    stream.  The step is therefore run a few times on the capture stream BEFORE the capture starts: during the
    capture every lookup hits the cache, so no host-to-device copy is recorded (a recorded copy would read its host
    source again at every replay) and no scratch buffer is allocated inside the graph;
  * tensors the step allocates while it is captured (its outputs) come from the graph's private memory pool and stay
    valid, at fixed addresses, for as long as the graph object lives: the outputs returned by ``capture`` are
    overwritten by every replay;
  * the step must not make the host wait for the GPU and must not read device values on the host: use the padded
    forms (``postprocess_batched(..., return_padded=True)``), not the per-image ``Detections`` lists;
  * shapes are frozen: the inputs of a replay are the tensors that were captured (copy new values into them), and the
    number of GT boxes per image is part of the recorded launch arguments;
  * the recorded launches hold raw addresses of those cached tables and scratch buffers, while the caches drop entries
    on their own schedule (the table cache is cleared when it is full -- a training loop whose GT counts change every
    step fills it --, a scratch buffer is replaced when a later call needs a larger one).  ``capture`` therefore
    collects every cached tensor the launchers hand out while the step is recorded and the ``CapturedStep`` keeps them
    alive for as long as the graph lives.
"""
from __future__ import annotations

from typing import Any, Callable, Sequence, Tuple

import torch

from . import _C, kernels


class CapturedStep:
    """A recorded step: ``replay()`` enqueues it on the current stream, ``outputs`` are its static output tensors."""

    def __init__(self, graph: "torch.cuda.CUDAGraph", outputs: Any, stream: "torch.cuda.Stream", held: Sequence[Any] = ()):
        self.graph = graph
        self.outputs = outputs
        self.capture_stream = stream
        self.held = tuple(held)        # cached tables / scratch buffers whose addresses the recorded launches use

    def replay(self) -> Any:
        self.graph.replay()
        return self.outputs


def capture(step: Callable[[], Any], device: torch.device, warmup: int = 3,
            side_streams: Sequence["torch.cuda.Stream"] = ()) -> CapturedStep:
    """Record ``step()`` (a function of no arguments that launches the work on the current stream and returns its
    output tensors) into a graph.  ``side_streams``: streams the step forks work onto and joins again itself (with
    ``wait_stream`` in both directions); they are warmed up with the capture stream so that their per-stream caches are
    filled as well."""
    _C.profile_select(_C.PROFILE_OFF)              # timed launches (events around a kernel) cannot be recorded
    stream = torch.cuda.Stream(device)
    stream.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(stream):
        for _ in range(max(1, warmup)):
            step()
    stream.synchronize()
    for s in side_streams:
        s.synchronize()
    graph = torch.cuda.CUDAGraph()
    previous, kernels._capture_refs = kernels._capture_refs, []
    try:
        with torch.cuda.graph(graph, stream=stream):
            outputs = step()
        held = kernels._capture_refs
    finally:
        kernels._capture_refs = previous
    return CapturedStep(graph, outputs, stream, held)
