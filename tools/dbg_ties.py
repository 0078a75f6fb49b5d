import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dataclasses, torch
from object_detection_amd import kernels, synthetic
from oracle import tv_reference as tv
dev = torch.device("cuda:0")
cfg = dataclasses.replace(synthetic.CFG2, batch=2, num_gt=20)
ag = tv.AnchorGenerator(synthetic.retinanet_anchor_sizes(len(cfg.grids)), ((0.5, 1.0, 2.0),) * len(cfg.grids))
grids = [(g, g) for g in cfg.grids]
strides = [(cfg.image_size[0] // g, cfg.image_size[1] // g) for g in cfg.grids]
anchors = kernels.anchor_generate(grids, strides, ag.cell_anchors, dev)
t = synthetic.make_targets(cfg)[0]["boxes"].to(dev)
iou = kernels.box_iou(t, anchors)
mx = iou.max(dim=1, keepdim=True).values
att = (iou == mx)
print("attaining per GT", att.sum(1).tolist())
print("max", mx.flatten().tolist())
idx = att.nonzero()
for g in range(3):
    a = idx[idx[:, 0] == g][:, 1]
    print(g, t[g].tolist(), a.tolist(), anchors[a].tolist())
