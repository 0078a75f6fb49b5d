"""Calibration: what a plain read stream of the filter kernel's input size achieves with torch kernels."""
import torch
dev = torch.device("cuda:0")
n = 16 * 786240 * 4
x = torch.randn(n, device=dev)
y = torch.empty_like(x)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for name, fn, bytes_ in (("sum", lambda: x.sum(), n * 4), ("max", lambda: x.max(), n * 4),
                         ("copy", lambda: y.copy_(x), n * 8), ("gt_count", lambda: (x > 3.0).sum(), n * 4)):
    us = timed(fn)
    print(f"{name}: {us:.1f} us  {bytes_ / us / 1e6:.2f} TB/s")
