"""``AnchorGenerator`` with the interface of ``torchvision.models.detection.anchor_utils`` (0.16.2),
as constructed by the reference at ``object_detection/retinanet/retinanet_model.py:22-27,42-46``.

The cell (base) anchors are tiny and computed on the host with the same fp32 torch ops torchvision
uses; the per-level grid expansion runs in one HIP launch and is cached per pyramid geometry, since
the result only depends on (grid sizes, strides, device).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
from torch import Tensor, nn

from . import kernels
from .image_list import ImageList


class AnchorGenerator(nn.Module):
    def __init__(self, sizes=((128, 256, 512),), aspect_ratios=((0.5, 1.0, 2.0),)):
        super().__init__()
        if not isinstance(sizes[0], (list, tuple)):
            sizes = tuple((s,) for s in sizes)
        if not isinstance(aspect_ratios[0], (list, tuple)):
            aspect_ratios = (aspect_ratios,) * len(sizes)
        self.sizes = sizes
        self.aspect_ratios = aspect_ratios
        self.cell_anchors = [self.generate_anchors(size, ratio) for size, ratio in zip(sizes, aspect_ratios)]
        self._cache: Dict[tuple, Tensor] = {}

    def generate_anchors(self, scales: Sequence[int], aspect_ratios: Sequence[float],
                         dtype: torch.dtype = torch.float32, device: torch.device = torch.device("cpu")) -> Tensor:
        # ratio-major, scale-minor; round() is half-to-even like torch.round
        scales = torch.as_tensor(scales, dtype=dtype, device=device)
        aspect_ratios = torch.as_tensor(aspect_ratios, dtype=dtype, device=device)
        h_ratios = torch.sqrt(aspect_ratios)
        w_ratios = 1 / h_ratios
        ws = (w_ratios[:, None] * scales[None, :]).view(-1)
        hs = (h_ratios[:, None] * scales[None, :]).view(-1)
        return (torch.stack([-ws, -hs, ws, hs], dim=1) / 2).round()

    def set_cell_anchors(self, dtype: torch.dtype, device: torch.device):
        self.cell_anchors = [c.to(dtype=dtype, device=device) for c in self.cell_anchors]

    def num_anchors_per_location(self) -> List[int]:
        return [len(s) * len(a) for s, a in zip(self.sizes, self.aspect_ratios)]

    def _grid_anchors_flat(self, grid_sizes, strides, device: torch.device = None):
        torch._assert(self.cell_anchors is not None, "cell_anchors should not be None")
        torch._assert(
            len(grid_sizes) == len(strides) == len(self.cell_anchors),
            "Anchors should be Tuple[Tuple[int]] because each feature "
            "map could potentially have different sizes and aspect ratios. "
            "There needs to be a match between the number of "
            "feature maps passed and the number of sizes / aspect ratios specified.",
        )
        device = device if device is not None else self.cell_anchors[0].device
        strides = [(int(s[0]), int(s[1])) for s in strides]
        grid_sizes = [(int(g[0]), int(g[1])) for g in grid_sizes]
        key = (tuple(grid_sizes), tuple(strides), device.type, device.index)
        flat = self._cache.get(key)
        if flat is None:
            flat = kernels.anchor_generate(grid_sizes, strides, self.cell_anchors, device)
            self._cache = {key: flat}      # one geometry at a time is enough (fixed-size training/inference)
        counts = [g[0] * g[1] * int(c.shape[0]) for g, c in zip(grid_sizes, self.cell_anchors)]
        return flat, counts

    def grid_anchors(self, grid_sizes: List[Tuple[int, int]], strides: List[Tuple[int, int]],
                     device: torch.device = None) -> List[Tensor]:
        flat, counts = self._grid_anchors_flat(grid_sizes, strides, device)
        return list(flat.split(counts))

    def forward(self, image_list: ImageList, feature_maps: List[Tensor]) -> List[Tensor]:
        grid_sizes = [tuple(int(v) for v in fm.shape[-2:]) for fm in feature_maps]
        image_size = image_list.tensors.shape[-2:]
        dtype, device = feature_maps[0].dtype, feature_maps[0].device
        strides = [(int(image_size[0]) // g[0], int(image_size[1]) // g[1]) for g in grid_sizes]
        all_levels, _ = self._grid_anchors_flat(grid_sizes, strides, device)
        if dtype != torch.float32:
            # torchvision adds int32 shifts to cell anchors already cast to the feature dtype; with
            # shifts below 2048 the fp32 sum is exact, so rounding it once gives the same half values
            max_shift = max(max(s[0] * (g[0] - 1), s[1] * (g[1] - 1)) for s, g in zip(strides, grid_sizes))
            if max_shift < 2048:
                all_levels = all_levels.to(dtype)
            else:
                parts = []
                for (gh, gw), (sh, sw), base in zip(grid_sizes, strides, self.cell_anchors):
                    sx = torch.arange(0, gw, dtype=torch.int32, device=device) * sw
                    sy = torch.arange(0, gh, dtype=torch.int32, device=device) * sh
                    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
                    sh4 = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
                    parts.append((sh4.view(-1, 1, 4) + base.to(dtype=dtype, device=device).view(1, -1, 4)).reshape(-1, 4))
                all_levels = torch.cat(parts)
        # the same tensor object is handed out for every image, as torchvision does
        return [all_levels for _ in range(len(image_list.image_sizes))]
