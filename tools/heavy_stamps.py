"""Time stamps of the cooperative matching kernel (a build with -DOD_HEAVY_STAMPS writes them over the first matched
indices of every heavy tile): tools/heavy_stamps.py LIB"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from object_detection_amd import _C, kernels, synthetic  # noqa: E402
from object_detection_amd.anchor_utils import AnchorGenerator  # noqa: E402

_C.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device("cuda:0")
cfg = synthetic.CFG2
ag = AnchorGenerator(synthetic.retinanet_anchor_sizes(len(cfg.grids)), ((0.5, 1.0, 2.0),) * len(cfg.grids))
grids = [(g, g) for g in cfg.grids]
strides = [(cfg.image_size[0] // g, cfg.image_size[1] // g) for g in cfg.grids]
anchors = kernels.anchor_generate(grids, strides, ag.cell_anchors, dev)
gts = [t["boxes"].to(dev) for t in synthetic.make_targets(cfg, empty_image=cfg.batch - 1)]
for _ in range(4):
    out = kernels.match_retinanet(anchors, gts, 0.6, 0.5, True)
torch.cuda.synchronize()
m = out[0].cpu()
pos = (m >= (1 << 40)).nonzero().flatten()
starts = pos[(m[pos] >> 40) == 1]
rows = []
for p in starts.tolist():
    v = m[p:p + 7]
    if all(int(v[i] >> 40) == i + 1 for i in range(7)):
        rows.append([int(v[i] & ((1 << 40) - 1)) / 100.0 for i in range(7)])
rows.sort(key=lambda r: -r[6])
print(len(rows), "heavy tiles in image 0; stamps (us): pick, anchors+bbox, GT loop, flush, merge barrier, end")
for r in rows[:6] + rows[-3:]:
    print(" ".join(f"{x:6.2f}" for x in r[1:]))
