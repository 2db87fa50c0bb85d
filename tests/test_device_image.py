"""The host half of the device data path (scda_amd/device_image.py): Pillow's resize coefficient tables, restated, applied with
plain integer numpy exactly as Pillow's 8-bit loops (and the HIP kernels) apply them -- against PIL itself."""
import numpy as np
import pytest

PIL = pytest.importorskip("PIL")
from PIL import Image

from scda_amd.device_image import PRECISION_BITS, axis_coeffs

PIL_FILTERS = {"bicubic": Image.BICUBIC, "bilinear": Image.BILINEAR, "box": Image.BOX, "hamming": Image.HAMMING, "lanczos": Image.LANCZOS}


def apply_tables(a, new_w, new_h, filter):
    """two passes, 8-bit intermediate, rows [row0, row0 + rows) only -- the kernels' arithmetic in numpy int64 (no overflow: the
    kernels' int32 sums stay below 2^31 like Pillow's)"""
    H, W, C = a.shape
    bh, kh, _ = axis_coeffs(W, new_w, filter)
    bv, kv, _ = axis_coeffs(H, new_h, filter)
    row0 = int(bv[0, 0]); rows = int(bv[-1, 0] + bv[-1, 1]) - row0
    src = a.astype(np.int64)
    half = 1 << (PRECISION_BITS - 1)
    tmp = np.zeros((rows, new_w, C), np.int64)
    for x in range(new_w):
        x0, n = bh[x]
        acc = half + (src[row0:row0 + rows, x0:x0 + n] * kh[x, :n].astype(np.int64)[None, :, None]).sum(1)
        assert np.abs(acc).max() < 2 ** 31
        tmp[:, x] = np.clip(acc >> PRECISION_BITS, 0, 255)
    out = np.zeros((new_h, new_w, C), np.uint8)
    for y in range(new_h):
        y0, n = bv[y]
        acc = half + (tmp[y0 - row0:y0 - row0 + n] * kv[y, :n].astype(np.int64)[:, None, None]).sum(0)
        assert np.abs(acc).max() < 2 ** 31
        out[y] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return out


SIZES = [(64, 128, 32, 64), (37, 91, 50, 120), (100, 60, 100, 33), (48, 48, 96, 48), (128, 256, 75, 150), (20, 30, 7, 11),
         (33, 65, 33, 65), (256, 512, 128, 256)]


@pytest.mark.parametrize("filter", sorted(PIL_FILTERS))
def test_tables_reproduce_pil_resize(filter):
    rng = np.random.default_rng(3)
    for H, W, nh, nw in SIZES:
        a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(a).resize((nw, nh), PIL_FILTERS[filter]))
        assert np.array_equal(apply_tables(a, nw, nh, filter), ref), (filter, H, W, nh, nw)


def test_default_filter_of_this_pillow_is_bicubic():
    # data.py calls img.resize(size) like the reference does (example_dataset.py:118): the device path must follow the default
    rng = np.random.default_rng(4)
    a = rng.integers(0, 256, (40, 72, 3), dtype=np.uint8)
    assert np.array_equal(np.asarray(Image.fromarray(a).resize((50, 31))), apply_tables(a, 50, 31, "bicubic"))


def test_extreme_values_and_single_channel():
    # saturated edges make the negative lobes of the cubic over- and undershoot: the clip to 0..255 is part of the arithmetic
    a = np.zeros((32, 48, 1), np.uint8); a[:, ::3] = 255; a[::5] = 255
    for nh, nw in ((21, 80), (64, 17)):
        ref = np.asarray(Image.fromarray(a[:, :, 0]).resize((nw, nh), Image.BICUBIC))
        assert np.array_equal(apply_tables(a, nw, nh, "bicubic")[:, :, 0], ref)


def test_identity_axis_is_exact():
    b, k, ks = axis_coeffs(17, 17)
    assert ks == 1 and (k == 1 << PRECISION_BITS).all() and (b[:, 0] == np.arange(17)).all() and (b[:, 1] == 1).all()
