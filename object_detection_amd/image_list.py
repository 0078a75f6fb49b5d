"""Batch container with the fields of ``torchvision.models.detection.image_list.ImageList``."""
from typing import List, Tuple

import torch
from torch import Tensor


class ImageList:
    """``tensors``: padded batch [B,C,H,W]; ``image_sizes``: (h, w) of each image before padding."""

    def __init__(self, tensors: Tensor, image_sizes: List[Tuple[int, int]]):
        self.tensors = tensors
        self.image_sizes = image_sizes

    def to(self, device: torch.device) -> "ImageList":
        return ImageList(self.tensors.to(device), self.image_sizes)
