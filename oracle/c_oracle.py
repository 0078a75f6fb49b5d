"""ctypes binding of oracle/libod_oracle.so (scalar C restatement) -- TEST INFRASTRUCTURE."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libod_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build() -> str:
    subprocess.check_call(["make", "-C", _HERE, "libod_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_nms.restype = ctypes.c_int64
        _lib.orc_batched_nms.restype = ctypes.c_int64
        _lib.orc_nms_rule.restype = ctypes.c_int64
        _lib.orc_batched_nms_rule.restype = ctypes.c_int64
        _lib.orc_filter_topk.restype = ctypes.c_int64
        _lib.orc_expf.restype = ctypes.c_float
        _lib.orc_expf.argtypes = [ctypes.c_float]
        _lib.orc_sigmoidf.restype = ctypes.c_float
        _lib.orc_sigmoidf.argtypes = [ctypes.c_float]
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(_i64p)


def expf(x):
    x, xp = _f(x)
    out = np.empty_like(x)
    lib().orc_expf_array(xp, out.ctypes.data_as(_f32p), ctypes.c_int64(x.size))
    return out


def sigmoidf(x):
    x, xp = _f(x)
    out = np.empty_like(x)
    lib().orc_sigmoidf_array(xp, out.ctypes.data_as(_f32p), ctypes.c_int64(x.size))
    return out


def box_iou(gt, anchors):
    gt, gp = _f(gt)
    anchors, ap = _f(anchors)
    m, n = gt.shape[0], anchors.shape[0]
    out = np.empty((m, n), dtype=np.float32)
    lib().orc_box_iou(gp, ctypes.c_int64(m), ap, ctypes.c_int64(n), out.ctypes.data_as(_f32p))
    return out


def match_retinanet(gt, anchors, high, low, allow_low_quality=True):
    gt, gp = _f(np.asarray(gt).reshape(-1, 4))
    anchors, ap = _f(anchors)
    m, n = gt.shape[0], anchors.shape[0]
    out = np.empty(n, dtype=np.int64)
    lib().orc_match_retinanet(gp, ctypes.c_int64(m), ap, ctypes.c_int64(n), ctypes.c_float(high),
                              ctypes.c_float(low), ctypes.c_int(int(allow_low_quality)),
                              out.ctypes.data_as(_i64p))
    return out


def matcher(q, high, low, allow_low_quality=True):
    q, qp = _f(q)
    m, n = q.shape
    out = np.empty(n, dtype=np.int64)
    lib().orc_matcher(qp, ctypes.c_int64(m), ctypes.c_int64(n), ctypes.c_float(high), ctypes.c_float(low),
                      ctypes.c_int(int(allow_low_quality)), out.ctypes.data_as(_i64p))
    return out


def match_fcos(gt, anchors, n_first_level, n_last_level, radius=1.5):
    gt, gp = _f(np.asarray(gt).reshape(-1, 4))
    anchors, ap = _f(anchors)
    m, n = gt.shape[0], anchors.shape[0]
    out = np.empty(n, dtype=np.int64)
    lib().orc_match_fcos(gp, ctypes.c_int64(m), ap, ctypes.c_int64(n), ctypes.c_int64(n_first_level),
                         ctypes.c_int64(n_last_level), ctypes.c_float(radius), out.ctypes.data_as(_i64p))
    return out


def nms(boxes, scores, thr, float_rule=False):
    """float_rule: compare the fp32 IoU with float(thr) (torchvision's CUDA kernel) instead of promoting the IoU to
    double (its CPU kernel)."""
    boxes, bp = _f(np.asarray(boxes).reshape(-1, 4))
    scores, sp = _f(scores)
    n = boxes.shape[0]
    keep = np.empty(max(n, 1), dtype=np.int64)
    nk = lib().orc_nms_rule(bp, sp, ctypes.c_int64(n), ctypes.c_double(thr), ctypes.c_int(int(float_rule)),
                            keep.ctypes.data_as(_i64p))
    return keep[:nk].copy()


def batched_nms(boxes, scores, labels, thr, mode, float_rule=False):
    """mode: 0 = coordinate trick, 1 = vanilla (per class)."""
    boxes, bp = _f(np.asarray(boxes).reshape(-1, 4))
    scores, sp = _f(scores)
    labels, lp = _i(labels)
    n = boxes.shape[0]
    keep = np.empty(max(n, 1), dtype=np.int64)
    nk = lib().orc_batched_nms_rule(bp, sp, lp, ctypes.c_int64(n), ctypes.c_double(thr), ctypes.c_int(mode),
                                    ctypes.c_int(int(float_rule)), keep.ctypes.data_as(_i64p))
    return keep[:nk].copy()


def decode_single(rel, anchors, clip):
    rel, rp = _f(np.asarray(rel).reshape(-1, 4))
    anchors, ap = _f(np.asarray(anchors).reshape(-1, 4))
    out = np.empty_like(rel)
    lib().orc_decode_single(rp, ap, ctypes.c_int64(rel.shape[0]), ctypes.c_float(clip),
                            out.ctypes.data_as(_f32p))
    return out


def decode_linear(rel, anchors):
    rel, rp = _f(np.asarray(rel).reshape(-1, 4))
    anchors, ap = _f(np.asarray(anchors).reshape(-1, 4))
    out = np.empty_like(rel)
    lib().orc_decode_linear(rp, ap, ctypes.c_int64(rel.shape[0]), out.ctypes.data_as(_f32p))
    return out


def clip(boxes, height, width):
    boxes = np.array(boxes, dtype=np.float32).reshape(-1, 4)
    lib().orc_clip(boxes.ctypes.data_as(_f32p), ctypes.c_int64(boxes.shape[0]), ctypes.c_float(height),
                   ctypes.c_float(width))
    return boxes


def filter_topk(logits, ctr, num_classes, thresh, k):
    logits, lp = _f(np.asarray(logits).reshape(-1))
    n_anchors = logits.size // num_classes
    if ctr is not None:
        ctr, cp = _f(np.asarray(ctr).reshape(-1))
    else:
        cp = None
    out_s = np.empty(max(k, 1), dtype=np.float32)
    out_i = np.empty(max(k, 1), dtype=np.int64)
    c = lib().orc_filter_topk(lp, cp, ctypes.c_int64(n_anchors), ctypes.c_int64(num_classes),
                              ctypes.c_float(thresh), ctypes.c_int64(k), out_s.ctypes.data_as(_f32p),
                              out_i.ctypes.data_as(_i64p))
    return out_s[:c].copy(), out_i[:c].copy()


def grid_anchors(base, gh, gw, sh, sw):
    base, bp = _f(np.asarray(base).reshape(-1, 4))
    A = base.shape[0]
    out = np.empty((gh * gw * A, 4), dtype=np.float32)
    lib().orc_grid_anchors(bp, ctypes.c_int64(A), ctypes.c_int64(gh), ctypes.c_int64(gw), ctypes.c_int64(sh),
                           ctypes.c_int64(sw), out.ctypes.data_as(_f32p))
    return out
