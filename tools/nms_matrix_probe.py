#!/usr/bin/env python3
"""Strip kernel against the chip-wide bit-matrix form of od_batched_nms (OD_NMS_PREFER_MATRIX): microseconds per call,
back to back on one stream, for a plain nms that keeps every survivor and for the pipeline's shape (sorted segments,
a few hundred kept)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd import _C, kernels  # noqa: E402

dev = torch.device("cuda:0")


def boxes_random(n, gen, extent, wh=80.0):
    xy = torch.rand((n, 2), generator=gen) * extent
    return torch.cat([xy, xy + torch.rand((n, 2), generator=gen) * wh + 1], dim=1)


def timed(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    gen = torch.Generator().manual_seed(1)
    print("plain nms, one image, unsorted, every survivor kept (threshold 0.5, CUDA rule)")
    for n, extent in ((300, 300.0), (600, 400.0), (1000, 500.0), (1500, 600.0), (4145, 1024.0), (4145, 4000.0), (8192, 2000.0)):
        b, s = boxes_random(n, gen, extent)[None].to(dev), torch.rand((1, n), generator=gen).to(dev)
        cnt = torch.tensor([[n]], dtype=torch.int32, device=dev)
        out = {}
        for name, flag in (("strips", 0), ("matrix", _C.NMS_PREFER_MATRIX)):
            mode = _C.NMS_TRICK | _C.NMS_THRESHOLD_FLOAT | flag
            call = lambda: kernels.batched_nms_padded(b, s, None, cnt, n, False, 0.5, mode, n, want_labels=False)
            kept = int(call()[4].item())
            out[name] = timed(call, 50 if n > 2000 else 200)
        print(f"  n = {n:5d}, extent {extent:6.0f}: kept {kept:5d}   strips {out['strips']:8.1f} us   matrix {out['matrix']:8.1f} us")
    print("pipeline shape: 6 sorted segments of 1000, 4 classes, NMS_AUTO_GPU, threshold 0.3")
    for B, D in ((16, 500), (1, 300)):
        n = 6000
        bx = torch.stack([boxes_random(n, gen, 1024.0, 120.0) for _ in range(B)]).to(dev)
        sc = torch.rand((B, 6, 1000), generator=gen).sort(dim=2, descending=True)[0].reshape(B, n).to(dev)
        lab = torch.randint(0, 4, (B, n), generator=gen, dtype=torch.int32).to(dev)
        cnt = torch.full((B, 6), 690, dtype=torch.int32, device=dev)          # 4140 candidates per image
        out = {}
        for name, flag in (("strips", 0), ("matrix", _C.NMS_PREFER_MATRIX)):
            mode = _C.NMS_AUTO_GPU | _C.NMS_THRESHOLD_FLOAT | flag
            call = lambda: kernels.batched_nms_padded(bx, sc, lab, cnt, 1000, True, 0.3, mode, D, want_keep_idx=False)
            kept = call()[4].tolist()
            out[name] = timed(call)
        print(f"  B = {B:2d}, D = {D}: kept {min(kept)}..{max(kept)}   strips {out['strips']:8.1f} us   matrix {out['matrix']:8.1f} us")


if __name__ == "__main__":
    main()
