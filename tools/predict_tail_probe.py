"""Duration of od_predict_tail (the prediction loop's rescale + class-agnostic NMS + int32 gather, reference
predict.py:70-76) on the detections of a cfg2 batch, against the same steps as torch launches per image.
Usage: python tools/predict_tail_probe.py [batch] [detections]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_amd import _C, kernels  # noqa: E402
if os.environ.get("OD_LIB"):
    _C.LIB_PATH = os.path.abspath(os.environ["OD_LIB"])

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
D = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(5)
xy = torch.rand((B, D, 2), generator=gen) * 960
boxes = torch.cat([xy, xy + torch.rand((B, D, 2), generator=gen) * 100 + 4], dim=2).to(dev)
scores = torch.rand((B, D), generator=gen).sort(dim=1, descending=True)[0].to(dev)
labels = torch.randint(1, 4, (B, D), generator=gen).to(dev)
counts = torch.full((B,), D, dtype=torch.int32, device=dev)
ratio = torch.tensor([[1920 / 1024, 1080 / 1024] * 2] * B, dtype=torch.float32, device=dev)
kid = _C.KERNEL_NAMES.index("predict_tail")
for _ in range(5):
    out = kernels.predict_tail(boxes.clone(), scores, labels, counts, ratio, 0.3)
torch.cuda.synchronize()
_C.profile_select(kid)
for _ in range(20):
    out = kernels.predict_tail(boxes.clone(), scores, labels, counts, ratio, 0.3)
torch.cuda.synchronize()
ms, n = _C.profile_read(kid)
_C.profile_select(_C.PROFILE_OFF)
t0 = time.perf_counter()
for _ in range(200):
    out = kernels.predict_tail(boxes, scores, labels, counts, ratio, 0.3, scale_in_place=False)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 200
print(f"od_predict_tail  B={B} D={D}: kernel {1e3 * ms / n:.1f} us, call incl. host {1e6 * wall:.1f} us, "
      f"kept {out[4].float().mean().item():.1f} of {D}")
