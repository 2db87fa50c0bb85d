"""The data path's per-image work on the device: resize + flip + ToTensor + Normalize of one decoded image.

The reference resizes with PIL (`img.resize((new_w, new_h))`, datasets/example_dataset.py:118, datasets/target_dataset.py:60-66)
and converts with torchvision's ToTensor / Normalize (faster_rcnn_train_val.py:195).  `Image.resize` is Pillow's `ImagingResample`
(src/libImaging/Resample.c of the Pillow the process imports -- a dependency of the reference, not under /root/reference; the
reference pins no version, this container has 12.2.0 whose default filter is BICUBIC): a two-pass separable convolution on 8-bit
samples in fixed point.  This module restates its coefficient set-up (`precompute_coeffs`, `normalize_coeffs_8bpc`) on the host,
in the same double-precision operations in the same order, and hands the integer tables to
`scda_image_resize_normalize_hip`, which applies them exactly as Pillow's 8-bit loops do.  Pinned against PIL itself:
tests/test_device_image.py (tables, through an integer numpy application) and tests/test_device_image_gpu.py (the kernels).
"""
import math

import numpy as np
import torch

from . import native as N

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _bilinear(x):
    if x < 0.0:
        x = -x
    if x < 1.0:
        return 1.0 - x
    return 0.0


def _box(x):
    return 1.0 if -0.5 < x <= 0.5 else 0.0


def _hamming(x):
    if x < 0.0:
        x = -x
    if x == 0.0:
        return 1.0
    if x >= 1.0:
        return 0.0
    x = x * math.pi
    return math.sin(x) / x * (0.54 + 0.46 * math.cos(x))


def _sinc(x):
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x):
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3)
    return 0.0


FILTERS = {"bicubic": (_bicubic, 2.0), "bilinear": (_bilinear, 1.0), "box": (_box, 0.5), "hamming": (_hamming, 1.0),
           "lanczos": (_lanczos, 3.0)}


def axis_coeffs(in_size, out_size, filter="bicubic"):
    """Pillow's per-axis tables for resizing `in_size` samples to `out_size` over the whole axis (box = (0, in_size)):
    bounds int32 [out_size, 2] = (first input sample, number of samples), kk int32 [out_size, ksize] fixed-point weights.
    Equal sizes: the identity (Pillow skips the pass; one weight of 2^22 reproduces the sample)."""
    if in_size == out_size:
        bounds = np.stack([np.arange(out_size, dtype=np.int32), np.ones(out_size, np.int32)], 1)
        return bounds, np.full((out_size, 1), 1 << PRECISION_BITS, np.int32), 1
    fn, fsupport = FILTERS[filter]
    scale = float(in_size) / out_size
    filterscale = scale if scale > 1.0 else 1.0
    support = fsupport * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.int32)
    bounds = np.zeros((out_size, 2), np.int32)
    ss = 1.0 / filterscale
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        k = []
        for x in range(xmax):
            w = fn((x + xmin - center + 0.5) * ss)
            k.append(w)
            ww += w
        for x in range(xmax):
            w = k[x] / ww if ww != 0.0 else k[x]
            kk[xx, x] = int(-0.5 + w * one) if w < 0 else int(0.5 + w * one)
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


_tables = {}


def resize_tables(in_h, in_w, out_h, out_w, device, filter="bicubic"):
    """device-resident tables of one (input size, output size) pair, cached: a data set has a handful of them"""
    key = (in_h, in_w, out_h, out_w, str(device), filter)
    t = _tables.get(key)
    if t is None:
        bh, kh, ksh = axis_coeffs(in_w, out_w, filter)
        bv, kv, ksv = axis_coeffs(in_h, out_h, filter)
        # the horizontal pass only produces the rows the vertical tables reach (Resample.c: ybox_first .. ybox_last)
        row0 = int(bv[0, 0])
        rows = int(bv[-1, 0] + bv[-1, 1]) - row0
        t = (N.upload(bh, device), N.upload(kh, device), ksh, N.upload(bv, device), N.upload(kv, device), ksv, row0, rows)
        _tables[key] = t
    return t


def resize_to_tensor(img, new_w, new_h, device, normalize=True, mean=0.5, std=0.5, flip=False, filter="bicubic"):
    """`img`: a PIL image (mode L or RGB) or a uint8 array [H, W] / [H, W, 1] / [H, W, 3]  ->  float32 [C, new_h, new_w] on `device`:
    normalize(to_tensor(img.resize((new_w, new_h)) [.transpose(FLIP_LEFT_RIGHT)])) of the CPU data path (scda_amd/data.py), bit for bit.
    Only the decoded bytes cross PCIe (a quarter of the float tensor, and before the down-scale or after it, whichever the caller holds)."""
    mode = getattr(img, "mode", None)
    if mode is not None:
        # PIL's Image.resize() default (what the CPU path calls) depends on the image and on Pillow: NEAREST for modes 'P' and '1' always
        # and for every mode before Pillow 7, BICUBIC otherwise; np.asarray of a 'P' image are palette indices.  This path restates the
        # BICUBIC / BILINEAR / ... convolution filters on L and RGB bytes only -- anything else must take the CPU path, loudly.
        if mode not in ("L", "RGB"):
            raise ValueError("device image path: PIL mode %r is not supported (L and RGB are); load with device=None or convert('RGB')" % mode)
        import PIL
        if int(PIL.__version__.split(".")[0]) < 7 and filter == "bicubic":
            raise ValueError("device image path: Pillow %s resizes with NEAREST by default; the device path restates Pillow >= 7's BICUBIC "
                             "default -- pass device=None" % PIL.__version__)
    a = np.asarray(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    H, W, C = a.shape
    src = N.upload(np.ascontiguousarray(a), device)
    return N.image_resize_normalize(src, resize_tables(H, W, new_h, new_w, device, filter), new_h, new_w, normalize, mean, std, flip)
