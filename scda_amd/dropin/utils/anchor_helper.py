"""Anchor grid in image coordinates -- the contract of utils/anchor_helper.py:21-35 (Detectron-style anchors).

Note the reference quirk that is preserved on purpose: `anchor_ratios` is accepted but NOT used; the grid is always
built with aspect ratios (0.5, 1, 2) (utils/anchor_helper.py:10-11 returns before the ratio code)."""
import numpy as np

_ASPECT = (0.5, 1.0, 2.0)


def _centred(ws, hs, cx, cy):
    ws = np.asarray(ws, dtype=np.float64).reshape(-1, 1)
    hs = np.asarray(hs, dtype=np.float64).reshape(-1, 1)
    return np.hstack([cx - 0.5 * (ws - 1), cy - 0.5 * (hs - 1), cx + 0.5 * (ws - 1), cy + 0.5 * (hs - 1)])


def generate_anchors(stride=16, sizes=(32, 64), aspect_ratios=_ASPECT):
    """[len(ratios)*len(sizes), 4] float64, ratio-major, centred on the stride cell (utils/anchor_helper.py:37-92)."""
    scales = np.array(sizes, dtype=np.float64) / stride
    ratios = np.array(aspect_ratios, dtype=np.float64)
    w = h = float(stride)
    ctr = 0.5 * (stride - 1)
    ws = np.round(np.sqrt(w * h / ratios))
    hs = np.round(ws * ratios)
    per_ratio = _centred(ws, hs, ctr, ctr)
    out = []
    for a in per_ratio:
        aw, ah = a[2] - a[0] + 1, a[3] - a[1] + 1
        acx, acy = a[0] + 0.5 * (aw - 1), a[1] + 0.5 * (ah - 1)
        out.append(_centred(aw * scales, ah * scales, acx, acy))
    return np.vstack(out)


def get_anchors_over_grid(ratios, scales, stride):
    return generate_anchors(stride=stride, sizes=np.array(scales) * stride)


_PLANE_CACHE = {}


def get_anchors_over_plane(featmap_h, featmap_w, anchor_ratios, anchor_scales, anchor_stride):
    """[K*A, 4] float64; row k*A + a is anchor a shifted to cell k = y*featmap_w + x.
    (The grid depends only on its arguments; it is built once per shape and handed out as a read-only array --
    the reference rebuilds it three times per iteration.)"""
    key = (featmap_h, featmap_w, tuple(anchor_scales), anchor_stride)
    hit = _PLANE_CACHE.get(key)
    if hit is not None:
        return hit
    out = _anchors_over_plane(featmap_h, featmap_w, anchor_ratios, anchor_scales, anchor_stride)
    out.setflags(write=False)
    _PLANE_CACHE[key] = out
    return out


def _anchors_over_plane(featmap_h, featmap_w, anchor_ratios, anchor_scales, anchor_stride):
    cell = get_anchors_over_grid(anchor_ratios, anchor_scales, anchor_stride)
    sx, sy = np.meshgrid(np.arange(featmap_w) * anchor_stride, np.arange(featmap_h) * anchor_stride)
    shifts = np.stack([sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel()], axis=1)
    return (cell[None, :, :] + shifts[:, None, :]).reshape(-1, 4)
