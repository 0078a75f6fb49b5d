"""Input/output transform around the box pipeline with the interface of torchvision's
``GeneralizedRCNNTransform`` (the reference sets ``min_size = max_size = size`` at
``object_detection/train.py:142-144``): normalise, resize so that the short side is ``min_size``
(capped by ``max_size``), pad the batch to a multiple of 32, rescale target boxes; ``postprocess``
maps detections back to the original image sizes.  Images on the GPU take one fused kernel
(``kernels.preprocess_images``: normalize + bilinear resize + zero-padded batch in a single pass); anything else
(CPU tensors, other dtypes, inputs that need a gradient) runs the same steps as PyTorch ops.
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
    ratios = _box_ratios(original_size, new_size)
    if boxes.is_cuda and boxes.dtype is torch.float32:
        return boxes * kernels._device_floats(ratios, boxes.device)         # one launch, no host->device copy
    xmin, ymin, xmax, ymax = boxes.unbind(1)
    return torch.stack((xmin * ratios[0], ymin * ratios[1], xmax * ratios[2], ymax * ratios[3]), dim=1)


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

    def normalize(self, image: Tensor) -> Tensor:
        if not image.is_floating_point():
            raise TypeError(f"Expected input images to be of floating type (in range [0, 1]), "
                            f"but found type {image.dtype} instead")
        mean = torch.as_tensor(self.image_mean, dtype=image.dtype, device=image.device)
        std = torch.as_tensor(self.image_std, dtype=image.dtype, device=image.device)
        return (image - mean[:, None, None]) / std[:, None, None]

    def resize(self, image: Tensor, target: Optional[Dict[str, Tensor]]):
        h, w = image.shape[-2:]
        size = float(self.min_size[-1])
        if self.fixed_size is not None:
            image = torch.nn.functional.interpolate(image[None], size=[self.fixed_size[1], self.fixed_size[0]],
                                                    mode="bilinear", align_corners=False)[0]
        else:
            scale = min(size / min(h, w), float(self.max_size) / max(h, w))
            image = torch.nn.functional.interpolate(image[None], scale_factor=scale, mode="bilinear",
                                                    recompute_scale_factor=True, align_corners=False)[0]
        if target is not None and "boxes" in target:
            target["boxes"] = resize_boxes(target["boxes"], (h, w), tuple(image.shape[-2:]))
        return image, target

    def batch_images(self, images: List[Tensor]) -> Tensor:
        max_size = [max(s) for s in zip(*[img.shape for img in images])]
        stride = float(self.size_divisible)
        max_size[1] = int(math.ceil(float(max_size[1]) / stride) * stride)
        max_size[2] = int(math.ceil(float(max_size[2]) / stride) * stride)
        batched = images[0].new_full([len(images)] + max_size, 0)
        for i, img in enumerate(images):
            batched[i, : img.shape[0], : img.shape[1], : img.shape[2]].copy_(img)
        return batched

    def _resized_size(self, h: int, w: int) -> Tuple[int, int]:
        """Size interpolate() gives an h x w image in ``resize``: floor(size * scale) in double arithmetic."""
        if self.fixed_size is not None:
            return int(self.fixed_size[1]), int(self.fixed_size[0])
        scale = min(float(self.min_size[-1]) / min(h, w), float(self.max_size) / max(h, w))
        return int(math.floor(float(h) * scale)), int(math.floor(float(w) * scale))

    def _forward_fused(self, images: List[Tensor], targets):
        sizes = [(int(img.shape[-2]), int(img.shape[-1])) for img in images]
        resized = [self._resized_size(h, w) for h, w in sizes]
        stride = float(self.size_divisible)
        out_h = int(math.ceil(float(max(r[0] for r in resized)) / stride) * stride)
        out_w = int(math.ceil(float(max(r[1] for r in resized)) / stride) * stride)
        if any(r[0] < 1 or r[1] < 1 for r in resized):
            return None
        batched = kernels.preprocess_images(images, resized, self.image_mean, self.image_std, (out_h, out_w))
        if targets is not None:
            for t, old, new in zip(targets, sizes, resized):
                if "boxes" in t:
                    t["boxes"] = resize_boxes(t["boxes"], old, new)
        return ImageList(batched, resized), targets

    def forward(self, images: List[Tensor], targets: Optional[List[Dict[str, Tensor]]] = None):
        images = [img for img in images]
        if targets is not None:
            targets = [{k: v for k, v in t.items()} for t in targets]   # do not modify the caller's dicts
        if (len(images) > 0 and all(img.dim() == 3 and img.is_cuda and img.dtype is torch.float32
                                    and not img.requires_grad and img.shape[0] == images[0].shape[0]
                                    for img in images)
                and images[0].shape[0] == len(self.image_mean) == len(self.image_std) and images[0].shape[0] <= 4
                and all(min(img.shape[-2:]) > 0 for img in images)):
            fused = self._forward_fused(images, targets)
            if fused is not None:
                return fused
        for i in range(len(images)):
            image = images[i]
            if image.dim() != 3:
                raise ValueError(f"images is expected to be a list of 3d tensors of shape [C, H, W], got {image.shape}")
            image = self.normalize(image)
            image, t = self.resize(image, targets[i] if targets is not None else None)
            images[i] = image
            if targets is not None:
                targets[i] = t
        image_sizes = [(int(img.shape[-2]), int(img.shape[-1])) for img in images]
        return ImageList(self.batch_images(images), image_sizes), targets

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
            result[i]["boxes"] = resize_boxes(pred["boxes"], im_s, o_im_s)
        return result
