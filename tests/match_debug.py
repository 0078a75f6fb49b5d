"""Debug helper: fused-path matching against the scalar checker on the unit-test pyramid; prints where they differ."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from object_detection_amd import kernels, synthetic  # noqa: E402
from oracle import c_oracle, tv_reference as tv  # noqa: E402


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda:0")
    grids = (64, 32, 16, 8, 4)
    sizes = synthetic.retinanet_anchor_sizes(5, 32)
    ag = tv.AnchorGenerator(sizes, ((0.5, 1.0, 2.0),) * 5)
    anchors = torch.cat(ag.grid_anchors([(g, g) for g in grids], [(512 // g, 512 // g) for g in grids]))
    gts = []
    for b in range(3):
        gen = torch.Generator().manual_seed(100 * m + b)
        gts.append(synthetic.make_gt_boxes(m if b != 1 else max(0, m - 1), (512, 512), gen))
    gts.append(torch.zeros((0, 4)))
    for lq in (True, False):
        out = kernels.match_retinanet(anchors.to(dev), [g.to(dev) for g in gts], 0.6, 0.5, lq).cpu()
        for b, g in enumerate(gts):
            ref = c_oracle.match_retinanet(g.numpy(), anchors.numpy(), 0.6, 0.5, lq)
            got = out[b].numpy()
            bad = np.nonzero(got != ref)[0]
            print(f"lq={lq} image {b}: {bad.size} differences", bad[:20].tolist(), got[bad[:10]].tolist(), ref[bad[:10]].tolist())


main()
