"""Checkpoint wire format of the reference: ``{'weights': state_dict, 'height': int, 'width': int}`` written by
``torch.save`` (reference object_detection/train.py:118-122) and read back by ``predict`` (reference
object_detection/predict.py:46-52: ``params['weights']``, ``params['height']``, ``params['width']``).

The detection heads of this package use torchvision's parameter names (``head.classification_head.conv.0.0.weight``,
``head.regression_head.bbox_reg.bias`` ...), so the head part of a reference checkpoint loads as it is; backbone
entries belong to the caller's backbone module.  Files are read with ``weights_only=True``: nothing in them is
executed.
"""
from __future__ import annotations

from typing import Dict, Iterable, Tuple

import torch
from torch import Tensor, nn


def save_checkpoint(model: nn.Module, path: str, height: int, width: int) -> None:
    torch.save({"weights": model.state_dict(), "height": int(height), "width": int(width)}, path)


def load_checkpoint(path: str, map_location="cpu") -> Dict[str, object]:
    params = torch.load(path, map_location=map_location, weights_only=True)
    if not isinstance(params, dict) or "weights" not in params:
        raise ValueError(f"{path}: not a {{'weights', 'height', 'width'}} checkpoint")
    return params


def load_weights(model: nn.Module, weights: Dict[str, Tensor], prefixes: Iterable[str] = ("head.",),
                 strict_shapes: bool = True) -> Tuple[list, list]:
    """Copy the entries of ``weights`` whose names start with one of ``prefixes`` into ``model`` (default: the
    detection heads).  Returns (loaded names, names the model has under those prefixes but the checkpoint lacks)."""
    own = model.state_dict()
    prefixes = tuple(prefixes)
    loaded, missing = [], []
    for name, dst in own.items():
        if not name.startswith(prefixes):
            continue
        src = weights.get(name)
        if src is None:
            missing.append(name)
            continue
        if tuple(src.shape) != tuple(dst.shape):
            if strict_shapes:
                raise ValueError(f"{name}: checkpoint has shape {tuple(src.shape)}, the model {tuple(dst.shape)}")
            missing.append(name)
            continue
        with torch.no_grad():
            dst.copy_(src)
        loaded.append(name)
    return loaded, missing
