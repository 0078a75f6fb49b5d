"""ctypes binding of ``libod_boxes.so`` (the C ABI declared in ``include/od_boxes.h``).

The library is the product: there is no CPU or eager fallback.  If the shared object is missing
or a symbol cannot be resolved, importing a kernel raises immediately with the build command.
"""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  -- must be loaded first: libod_boxes.so has to bind to the HIP runtime torch uses
from ctypes import c_double, c_float, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libod_boxes.so")

OD_MAX_LEVELS = 8
OD_MAX_BASE_ANCHORS = 16
OD_MAX_TOPK = 8192
OD_MAX_NMS_BOXES = 65536
OD_NMS_NOALLOC_SCRATCH_BYTES = 8 << 20
OD_MAX_PREDICT_TAIL = 1024
CODER_BOXCODER, CODER_LINEAR = 0, 1
NMS_TRICK, NMS_VANILLA, NMS_AUTO_CPU, NMS_AUTO_GPU = 0, 1, 2, 3
NMS_PREFER_MATRIX = 0x200       # or-ed into the mode: the chip-wide bit-matrix form when the call qualifies (see od_boxes.h)
NMS_MATRIX_MAX_BOXES = 8192
NMS_THRESHOLD_FLOAT = 0x100     # or-ed into the mode: compare the fp32 IoU with float(threshold), as torchvision's CUDA kernel

_ERRORS = {
    -1: "invalid argument (null pointer, negative size, misaligned or inconsistent shapes)",
    -2: "size beyond a documented limit of libod_boxes",
    -3: "workspace too small",
}

_P = c_void_p
# name -> (restype, argtypes); must list every symbol of include/od_boxes.h
SIGNATURES = {
    "od_version": (c_int32, []),
    "od_anchor_generate": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, _P, c_int64, _P]),
    "od_match_retinanet_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int64]),
    "od_match_retinanet": (c_int32, [_P, c_int64, c_int64, _P, _P, c_int64, c_int32, c_float, c_float, c_int32,
                                     _P, _P, c_size_t, _P]),
    "od_match_retinanet_grids": (c_int32, [_P, c_int64, c_int64, _P, c_int32, _P, _P, c_int64, c_int32, c_float, c_float,
                                           c_int32, _P, _P, c_size_t, _P]),
    "od_match_fcos": (c_int32, [_P, c_int64, c_int64, c_int64, c_int64, _P, _P, c_int32, c_float, _P, _P]),
    "od_box_iou": (c_int32, [_P, c_int64, _P, c_int64, _P, _P]),
    "od_matcher_workspace_bytes": (c_size_t, [c_int64, c_int64]),
    "od_matcher": (c_int32, [_P, c_int64, c_int64, c_float, c_float, c_int32, _P, _P, c_size_t, _P]),
    "od_decode_boxes": (c_int32, [_P, _P, c_int64, c_int32, c_float, c_float, c_float, _P, _P]),
    "od_select_candidates_workspace_bytes": (c_size_t, [_P, c_int32, c_int32, c_int32]),
    "od_select_candidates": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, _P, c_float,
                                       c_int32, c_int32, c_float, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "od_select_candidates_planar": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, _P,
                                              c_float, c_int32, c_int32, c_float, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "od_batched_nms_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "od_batched_nms_matrix_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "od_batched_nms": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_double, c_int32, c_int32,
                                 _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "od_batched_nms_noalloc": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_double, c_int32, c_int32,
                                         _P, _P, _P, _P, _P, _P]),
    "od_pack_detections": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, _P, _P]),
    "od_predict_tail_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "od_predict_tail": (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32, c_double, c_int32, _P, _P, _P, _P, _P, _P, _P,
                                  c_size_t, _P]),
    "od_focal_loss_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int32]),
    "od_focal_loss_forward": (c_int32, [_P, c_int32, c_int64, c_int64, c_int32, _P, _P, _P, c_int32, c_float, c_float,
                                        c_int32, _P, _P, _P, _P, c_size_t, _P]),
    "od_focal_loss_backward": (c_int32, [_P, c_int32, c_int64, c_int64, c_int32, _P, _P, _P, c_int32, c_float, c_float,
                                         c_int32, _P, _P, _P, _P]),
    "od_fcos_box_loss_forward": (c_int32, [_P, _P, c_int32, c_int64, c_int64, _P, c_int64, c_int64, _P, _P, _P, c_int32,
                                           c_float, _P, _P, _P, _P, c_size_t, _P]),
    "od_fcos_box_loss_backward": (c_int32, [_P, _P, c_int32, c_int64, c_int64, _P, c_int64, c_int64, _P, _P, _P, c_int32,
                                            c_float, _P, _P, _P, _P, _P, _P]),
    "od_box_loss_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "od_box_loss_forward": (c_int32, [_P, c_int32, c_int64, _P, c_int64, c_int64, _P, _P, _P, c_int32, c_int32, c_float,
                                      _P, c_float, c_float, _P, _P, _P, _P, c_size_t, _P]),
    "od_box_loss_backward": (c_int32, [_P, c_int32, c_int64, _P, c_int64, c_int64, _P, _P, _P, c_int32, c_int32, c_float,
                                       _P, c_float, c_float, _P, _P, _P, _P]),
    "od_preprocess_images": (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32, _P, _P, _P, c_int32, c_int32, _P]),
    "od_match_restore_groups": (c_int32, [c_int32]),
    "od_profile_select": (c_int32, [c_int32]),
    "od_profile_stride": (c_int32, [c_int32]),
    "od_profile_read": (c_int32, [c_int32, _P, _P]),
}

PROFILE_OFF, PROFILE_ALL = -1, -2
KERNEL_NAMES = ["anchor", "match_assign", "match_restore", "match_fcos", "box_iou", "matcher", "decode", "filter",
                "select", "nms", "match_heavy", "match_init", "focal_fwd", "focal_bwd", "box_loss_fwd", "box_loss_bwd", "preprocess", "sample", "filter_redo",
                "pack", "nms_order", "predict_tail"]
BOX_LOSS_L1, BOX_LOSS_SMOOTH_L1, BOX_LOSS_GIOU = 0, 1, 2
DTYPE_F32, DTYPE_F16, DTYPE_BF16 = 0, 1, 2
LOSS_NORM_PER_IMAGE, LOSS_NORM_BATCH = 0, 1

_lib = None


class OdBoxesError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load libod_boxes.so once; raise loudly if it is absent (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP kernels are not built. Run "
                f"`make -C {os.path.join(_HERE, 'csrc')}` (needs hipcc, targets gfx950) or "
                f"`python -c 'import __graft_entry__ as g; g.build()'` from the repository root.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(code: int, what: str) -> None:
    if code == 0:
        return
    if code < 0:
        raise OdBoxesError(f"{what}: {_ERRORS.get(code, 'error')} (code {code})")
    raise OdBoxesError(f"{what}: HIP runtime error {code}")


def match_restore_groups(groups: int) -> int:
    """Workgroups per image of the matching path's restore kernel (0: the default); returns the previous setting.
    Few (3) when matching runs beside other GPU work, the default (16) when the GPU waits for it."""
    prev = lib().od_match_restore_groups(int(groups))
    if prev < 0:
        check(prev, "od_match_restore_groups")
    return prev


def profile_select(kernel_id: int) -> None:
    check(lib().od_profile_select(int(kernel_id)), "od_profile_select")


def profile_stride(every: int) -> None:
    """Time only every ``every``-th launch of the selected kernel(s)."""
    check(lib().od_profile_stride(int(every)), "od_profile_stride")


def profile_read(kernel_id: int):
    """(total milliseconds, launches) of one kernel since the previous read; waits for its events."""
    ms, n = c_double(0.0), c_int64(0)
    check(lib().od_profile_read(int(kernel_id), ctypes.byref(ms), ctypes.byref(n)), "od_profile_read")
    return ms.value, n.value
