import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from object_detection_amd import _C, kernels
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(1)
def boxes_random(n, extent, wh=120.0):
    xy = torch.rand((n, 2), generator=gen) * extent
    return torch.cat([xy, xy + torch.rand((n, 2), generator=gen) * wh + 1], dim=1)
for B, nseg, stride, cntv, classes, D, thr in ((16, 6, 1000, 690, 4, 500, 0.3), (1, 6, 1000, 690, 4, 300, 0.3), (16, 5, 1000, 800, 4, 100, 0.6), (1, 6, 1000, 1000, 91, 100, 0.5), (16, 6, 1000, 690, 4, 200, 0.3), (16, 6, 1000, 690, 4, 100, 0.3)):
    n = nseg * stride
    bx = torch.stack([boxes_random(n, 1024.0) for _ in range(B)]).to(dev)
    sc = torch.rand((B, nseg, stride), generator=gen).sort(dim=2, descending=True)[0].reshape(B, n).to(dev)
    lab = torch.randint(0, classes, (B, n), generator=gen, dtype=torch.int32).to(dev)
    cnt = torch.full((B, nseg), cntv, dtype=torch.int32, device=dev)
    call = lambda: kernels.batched_nms_padded(bx, sc, lab, cnt, stride, True, thr, _C.NMS_AUTO_GPU | _C.NMS_THRESHOLD_FLOAT, D, want_keep_idx=False)
    for _ in range(10): call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): call()
    torch.cuda.synchronize()
    print(f"B={B} segs={nseg}x{cntv} classes={classes} D={D}: {(time.perf_counter()-t0)/300*1e6:.1f} us", flush=True)
