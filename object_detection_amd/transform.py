"""Input/output transform around the box pipeline with the interface of torchvision's
``GeneralizedRCNNTransform`` (the reference sets ``min_size = max_size = size`` at
``object_detection/train.py:142-144``): normalise, resize so that the short side is ``min_size``
(capped by ``max_size``), pad the batch to a multiple of 32, rescale target boxes; ``postprocess``
maps detections back to the original image sizes.  The image side is one kernel
(``kernels.preprocess_images``: normalize + bilinear resize + zero-padded batch in a single pass) for fp32 images on
the GPU; there is no second implementation, other inputs are rejected.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor, nn

from . import kernels
from .image_list import ImageList


def _box_ratios(original_size: Tuple[int, int], new_size: Tuple[int, int]) -> Tuple[float, float, float, float]:
    """(ratio_w, ratio_h, ratio_w, ratio_h) with ratio = fp32(new) / fp32(original), the fp32 quotient torchvision's
    resize_boxes forms on the device, computed on the host."""
    rh, rw = [float(np.float32(s) / np.float32(s_orig)) for s, s_orig in zip(new_size, original_size)]
    return rw, rh, rw, rh


def resize_boxes(boxes: Tensor, original_size: Tuple[int, int], new_size: Tuple[int, int]) -> Tensor:
    """Boxes scaled by the fp32 ratios (integer boxes -- the reference's dataset yields int64 -- become fp32, as the
    multiplication by a float tensor does in torchvision); one multiply by a cached row of four factors."""
    return boxes * kernels._device_floats(_box_ratios(original_size, new_size), boxes.device)


class GeneralizedRCNNTransform(nn.Module):
    def __init__(self, min_size, max_size, image_mean, image_std, size_divisible: int = 32,
                 fixed_size: Optional[Tuple[int, int]] = None, **kwargs):
        # unknown keyword arguments are accepted and ignored: the reference passes a misspelt
        # `nms_thres=` (object_detection/train.py:144) that ends up here
        super().__init__()
        self.min_size = (min_size,) if not isinstance(min_size, (list, tuple)) else tuple(min_size)
        self.max_size = max_size
        self.image_mean = image_mean
        self.image_std = image_std
        self.size_divisible = size_divisible
        self.fixed_size = fixed_size

    def _resized_size(self, h: int, w: int) -> Tuple[int, int]:
        """Size torchvision's ``resize`` gives an h x w image in eager mode (what the reference's train.py and
        predict.py run): the scale factor is the smaller of ``min_size / short side`` and ``max_size / long side`` in
        Python floats (doubles) -- only the scripted / traced branch of ``_resize_image_and_masks`` forms it from fp32
        tensors.  ``interpolate(recompute_scale_factor=True)`` then takes floor(size * scale) in double arithmetic:
        64 / 100 keeps 64 rows and 97 / 97 keeps 97."""
        if self.fixed_size is not None:
            return int(self.fixed_size[1]), int(self.fixed_size[0])
        scale = min(float(self.min_size[-1]) / float(min(h, w)), float(self.max_size) / float(max(h, w)))
        return int(math.floor(float(h) * scale)), int(math.floor(float(w) * scale))

    def forward(self, images: List[Tensor], targets: Optional[List[Dict[str, Tensor]]] = None):
        images = [img for img in images]
        if targets is not None:
            targets = [{k: v for k, v in t.items()} for t in targets]   # do not modify the caller's dicts
        if len(images) == 0:
            raise ValueError("images is expected to be a non-empty list of 3d tensors of shape [C, H, W]")
        for img in images:
            if img.dim() != 3:
                raise ValueError(f"images is expected to be a list of 3d tensors of shape [C, H, W], got {img.shape}")
            if not img.is_floating_point():
                raise TypeError(f"Expected input images to be of floating type (in range [0, 1]), "
                                f"but found type {img.dtype} instead")
            kernels._require_gpu(img, "images")
        channels = int(images[0].shape[0])
        if not (all(img.dtype is torch.float32 and not img.requires_grad and img.shape[0] == channels
                    and min(img.shape[-2:]) > 0 for img in images)
                and channels == len(self.image_mean) == len(self.image_std) and channels <= 4):
            raise RuntimeError("the input transform takes fp32 images of one channel count (at most 4, matching "
                               "image_mean / image_std) that need no gradient")
        sizes = [(int(img.shape[-2]), int(img.shape[-1])) for img in images]
        resized = [self._resized_size(h, w) for h, w in sizes]
        if any(r[0] < 1 or r[1] < 1 for r in resized):
            raise ValueError(f"an image would be resized to an empty image: {sizes} -> {resized}")
        stride = float(self.size_divisible)
        out_h = int(math.ceil(float(max(r[0] for r in resized)) / stride) * stride)
        out_w = int(math.ceil(float(max(r[1] for r in resized)) / stride) * stride)
        batched = kernels.preprocess_images(images, resized, self.image_mean, self.image_std, (out_h, out_w))
        if targets is not None and all("boxes" in t for t in targets):
            # resize_boxes for the whole batch: the boxes of all images in one buffer (two launches instead of one per
            # image), scaled by each image's ratios, handed back as per-image views -- which the matching and loss
            # kernels then read in place
            boxes = [t["boxes"] for t in targets]
            if all(b.is_cuda and b.dim() == 2 and b.shape[-1] == 4 for b in boxes) and sum(int(b.shape[0]) for b in boxes) > 0:
                views = kernels.pack_boxes(boxes)
                packed = torch.as_strided(views[0], (sum(int(v.shape[0]) for v in views), 4), (4, 1), views[0].storage_offset())
                flat = [v for old, new in zip(sizes, resized) for v in _box_ratios(old, new)]
                ratios = kernels._device_floats(flat, packed.device).view(-1, 4)
                counts = kernels._device_ints([int(v.shape[0]) for v in views], packed.device, torch.int64)
                packed.mul_(torch.repeat_interleave(ratios, counts, dim=0, output_size=packed.shape[0]))
                for t, v in zip(targets, views):
                    t["boxes"] = v
            else:
                for t, old, new in zip(targets, sizes, resized):
                    t["boxes"] = resize_boxes(t["boxes"], old, new)
        elif targets is not None:
            for t, old, new in zip(targets, sizes, resized):
                if "boxes" in t:
                    t["boxes"] = resize_boxes(t["boxes"], old, new)
        return ImageList(batched, resized), targets

    def box_scales(self, image_shapes: List[Tuple[int, int]], original_image_sizes: List[Tuple[int, int]],
                   device: torch.device) -> Tensor:
        """[B, 4] factors that ``postprocess`` applies to the boxes of each image (one multiply for a padded batch)."""
        flat = [v for im_s, o_im_s in zip(image_shapes, original_image_sizes) for v in _box_ratios(im_s, o_im_s)]
        return kernels._device_floats(flat, device).view(-1, 4)

    def postprocess(self, result: List[Dict[str, Tensor]], image_shapes: List[Tuple[int, int]],
                    original_image_sizes: List[Tuple[int, int]]) -> List[Dict[str, Tensor]]:
        if self.training:
            return result
        for i, (pred, im_s, o_im_s) in enumerate(zip(result, image_shapes, original_image_sizes)):
            pending = getattr(pred, "_pending", None)            # detections that are not cut to size yet
            if pending is not None and pred.defer_box_scale(
                    kernels._device_floats(_box_ratios(im_s, o_im_s), pending.boxes.device)):
                continue
            result[i]["boxes"] = resize_boxes(pred["boxes"], im_s, o_im_s)
        return result
