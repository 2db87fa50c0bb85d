"""numpy front-end of oracle/liboracle.so (scda_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Each function mirrors one C restatement; see scda_oracle.c for the reference
file:line citations.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


F = ctypes.c_float
I = ctypes.c_int


def nms(boxes, thresh):
    boxes = _f32(boxes)
    n = boxes.shape[0]
    keep = np.zeros(max(n, 1), dtype=np.int64)
    num = np.zeros(1, dtype=np.int64)
    lib().orc_nms(_p(boxes), I(n), F(thresh), _p(keep), _p(num))
    return keep[: int(num[0])].copy()


def nms_mask(boxes, thresh):
    boxes = _f32(boxes)
    n = boxes.shape[0]
    mask = np.zeros((n, (n + 63) // 64), dtype=np.uint64)
    if n:
        lib().orc_nms_mask(_p(boxes), I(n), F(thresh), _p(mask))
    return mask


def cpu_nms(boxes, order, areas, thresh):
    boxes = _f32(boxes)
    order = np.ascontiguousarray(order, dtype=np.int64)
    areas = _f32(areas)
    n = boxes.shape[0]
    keep = np.zeros(max(n, 1), dtype=np.int64)
    num = np.zeros(1, dtype=np.int64)
    lib().orc_cpu_nms(_p(boxes), I(n), I(boxes.shape[1]), _p(order), _p(areas), F(thresh), _p(keep), _p(num))
    return keep[: int(num[0])].copy()


def roi_pool_fwd(feat, rois, ph, pw, scale):
    feat = _f32(feat); rois = _f32(rois)
    B, C, H, W = feat.shape
    R = rois.shape[0]
    out = np.zeros((R, C, ph, pw), dtype=np.float32)
    arg = np.zeros((R, C, ph, pw), dtype=np.int32)
    lib().orc_roi_pool_fwd(_p(feat), _p(rois), I(R), I(C), I(H), I(W), I(ph), I(pw), F(scale), _p(out), _p(arg))
    return out, arg


def roi_pool_bwd(top, arg, rois, feat_shape, ph, pw, scale):
    top = _f32(top); rois = _f32(rois)
    arg = np.ascontiguousarray(arg, dtype=np.int32)
    B, C, H, W = feat_shape
    g = np.zeros((B, C, H, W), dtype=np.float32)
    lib().orc_roi_pool_bwd(_p(top), _p(arg), _p(rois), I(rois.shape[0]), I(B), I(C), I(H), I(W), I(ph), I(pw), F(scale),
                           _p(g))
    return g


def roi_pool_bwd_scatter(top, arg, feat_shape, ph, pw):
    top = _f32(top)
    arg = np.ascontiguousarray(arg, dtype=np.int32)
    B, C, H, W = feat_shape
    g = np.zeros((B, C, H, W), dtype=np.float32)
    lib().orc_roi_pool_bwd_scatter(_p(top), _p(arg), I(top.shape[0]), I(B), I(C), I(H), I(W), I(ph), I(pw), _p(g))
    return g


def roi_align_fwd(feat, rois, ah, aw, scale):
    feat = _f32(feat); rois = _f32(rois)
    B, C, H, W = feat.shape
    R = rois.shape[0]
    out = np.zeros((R, C, ah, aw), dtype=np.float32)
    lib().orc_roi_align_fwd(_p(feat), _p(rois), I(R), I(C), I(H), I(W), I(ah), I(aw), F(scale), _p(out))
    return out


def roi_align_bwd(top, rois, feat_shape, ah, aw, scale):
    top = _f32(top); rois = _f32(rois)
    B, C, H, W = feat_shape
    g = np.zeros((B, C, H, W), dtype=np.float32)
    lib().orc_roi_align_bwd(_p(top), _p(rois), I(rois.shape[0]), I(C), I(H), I(W), I(ah), I(aw), F(scale), _p(g))
    return g


def focal_sigmoid_fwd(logits, targets, weight_pos, gamma, alpha, nc):
    logits = _f32(logits); targets = np.ascontiguousarray(targets, dtype=np.int32)
    out = np.zeros_like(logits)
    lib().orc_focal_sigmoid_fwd(I(logits.size), _p(logits), _p(targets), F(weight_pos), F(gamma), F(alpha), I(nc), _p(out))
    return out


def focal_sigmoid_bwd(logits, targets, weight_pos, gamma, alpha, nc):
    logits = _f32(logits); targets = np.ascontiguousarray(targets, dtype=np.int32)
    out = np.zeros_like(logits)
    lib().orc_focal_sigmoid_bwd(I(logits.size), _p(logits), _p(targets), _p(out), F(weight_pos), F(gamma), F(alpha), I(nc))
    return out


def focal_softmax_fwd(logits, targets, weight_pos, gamma, alpha, nc):
    logits = _f32(logits); targets = np.ascontiguousarray(targets, dtype=np.int32)
    losses = np.zeros(logits.size // nc, dtype=np.float32)
    priors = np.zeros_like(logits)
    lib().orc_focal_softmax_fwd(I(logits.size), _p(logits), _p(targets), F(weight_pos), F(gamma), F(alpha), I(nc),
                                _p(losses), _p(priors))
    return losses, priors


def focal_softmax_bwd(logits, targets, priors, weight_pos, gamma, alpha, nc):
    logits = _f32(logits); targets = np.ascontiguousarray(targets, dtype=np.int32); priors = _f32(priors)
    dx = np.zeros_like(logits)
    buff = np.zeros(logits.size // nc, dtype=np.float32)
    lib().orc_focal_softmax_bwd(I(logits.size), _p(logits), _p(targets), _p(dx), F(weight_pos), F(gamma), F(alpha), I(nc),
                                _p(priors), _p(buff))
    return dx


def iou_overlaps(b1, b2):
    b1 = _f32(b1); b2 = _f32(b2)
    out = np.zeros((b1.shape[0], b2.shape[0]), dtype=np.float32)
    lib().orc_iou_overlaps(_p(b1), _p(b2), I(b1.shape[1]), I(b1.shape[0]), I(b2.shape[0]), _p(out))
    return out


def bbox_overlaps(boxes, query):
    boxes = _f32(boxes); query = _f32(query)
    out = np.zeros((boxes.shape[0], query.shape[0]), dtype=np.float32)
    lib().orc_bbox_overlaps(_p(boxes), I(boxes.shape[0]), _p(query), I(query.shape[0]), _p(out))
    return out
