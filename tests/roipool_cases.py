"""Seeded RoI max-pool inputs shared by tests/golden/make_golden_roipool.py (which runs the REFERENCE's own
extensions/_roi_pooling/modules/roi_pool_py.py on them) and by the tests that compare the oracle and the HIP kernel with what
it produced.  Only RandomState.uniform / randint / standard_normal and IEEE arithmetic: bit-identical arrays wherever they are
regenerated; the fixture stores a sha256 of every input, checked before use.

Two properties of every case, because roi_pool_py.py and the CUDA kernel that is the canonical text for the hot path
(roi_pooling_kernel.cu:24-93) are two statements of the same operator that part ways in exactly two places:
  * rounding of the scaled corner: np.round is half-to-even (roi_pool_py.py:21), CUDA round() is half-away-from-zero (:45-48).
    No scaled coordinate of any case lies on an exact .5  (`on_half`).
  * bin edges: roi_pool_py.py:25-35 computes size / pooled and (p + 1) * bin in Python doubles, the kernel in fp32 (:54-61).
    For some integer sizes the two disagree by one cell in ceil((p + 1) * bin) (e.g. size 29 over 7 bins: 29 / 7 * 7 is
    29.000000000000004 in double and 29.0f in float).  No RoI of any case has such a width or height (`edge_divergent`).
The two excluded situations are covered by tests of their own against the kernel's text
(tests/test_oracle_golden.py::test_roi_pool_half_away_rounding, ::test_roi_pool_fp32_bin_edges)."""
import hashlib

import numpy as np

# (name, feature shape [B,C,H,W], R, pooled_h, pooled_w, spatial_scale)
CASES = [
    ("full_512x32x64_r512", (1, 512, 32, 64), 512, 7, 7, 1.0 / 16),     # the hot path's call: BASELINE configs[1]
    ("ragged_b2_6x11x19_r40", (2, 6, 11, 19), 40, 7, 7, 1.0 / 16),       # two images, odd map, RoIs partly / fully outside
    ("ragged_3x40x9_r33_p3x5", (1, 3, 40, 9), 33, 3, 5, 1.0 / 8),        # non-square pooling, tall map, ties in the features
]


def _case(name):
    return next(c for c in CASES if c[0] == name)


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def on_half(v):
    """scaled coordinate(s) (float32) sitting exactly on x.5"""
    v = np.asarray(v, np.float32)
    return (v - np.floor(v)) == np.float32(0.5)


def edge_divergent(size, pooled):
    """does ANY bin edge of an integer RoI extent differ between double (roi_pool_py.py) and float (CUDA) arithmetic?"""
    b64 = float(size) / float(pooled)
    b32 = np.float32(size) / np.float32(pooled)
    for p in range(pooled):
        if int(np.floor(p * b64)) != int(np.floor(np.float32(p) * b32)):
            return True
        if int(np.ceil((p + 1) * b64)) != int(np.ceil(np.float32(p + 1) * b32)):
            return True
    return False


def rect(roi, scale):
    """integer rectangle of one RoI row on the map, half-away rounding in float32 (both statements agree off the .5 points)"""
    v = roi[1:5].astype(np.float32) * np.float32(scale)
    r = np.where(v >= 0, np.floor(v + np.float32(0.5)), np.ceil(v - np.float32(0.5))).astype(np.int64)
    return int(r[0]), int(r[1]), int(r[2]), int(r[3])


def make(name):
    """-> (features float32 [B,C,H,W], rois float32 [R,5] = (batch, x1, y1, x2, y2) in image coordinates)"""
    _, shape, R, PH, PW, scale = _case(name)
    B, C, H, W = shape
    rs = np.random.RandomState(int(hashlib.sha256(name.encode()).hexdigest()[:8], 16))
    feat = rs.standard_normal(shape).astype(np.float32)
    if name.startswith("ragged_3x40x9"):
        feat = np.round(feat * 2) / 2                        # few distinct values: every bin holds ties (first maximum wins)
        feat = feat.astype(np.float32) + np.float32(0.0)       # no -0.0: max over {-0.0, +0.0} is a tie whose SIGN the two statements may pick differently
    iw, ih = W / scale, H / scale                            # image extent
    # three populations, as the detector produces them: regression output (real-valued, clipped to the image), ground-truth
    # boxes (integers), and boxes reaching past the borders / lying outside (the kernel clips bins, empty bins give 0)
    x1 = rs.uniform(0, iw - 4, R); y1 = rs.uniform(0, ih - 4, R)
    w = np.exp(rs.uniform(np.log(2.0), np.log(iw * 0.7), R)); h = np.exp(rs.uniform(np.log(2.0), np.log(ih * 0.9), R))
    rois = np.stack([rs.randint(0, B, R).astype(np.float64), x1, y1, np.minimum(x1 + w, iw - 1), np.minimum(y1 + h, ih - 1)], 1)
    third = R // 3
    rois[third:2 * third, 1:] = np.round(rois[third:2 * third, 1:])
    far = np.arange(R - R // 6, R)
    rois[far, 1] += rs.uniform(-0.2, 0.25, len(far)) * iw
    rois[far, 3] += rs.uniform(-0.2, 0.25, len(far)) * iw
    rois[far, 2] += rs.uniform(-0.2, 0.25, len(far)) * ih
    rois[far, 4] += rs.uniform(-0.2, 0.25, len(far)) * ih
    rois[far, 3] = np.maximum(rois[far, 3], rois[far, 1]); rois[far, 4] = np.maximum(rois[far, 4], rois[far, 2])
    rois[far[0], 1:] = [iw + 40, 8, iw + 90, 30]             # entirely right of the map: every bin empty
    rois[far[1], 1:] = [3, 3, 3, 3]                          # one cell
    rois = rois.astype(np.float32)
    step = np.float32(0.25 / scale / 4)                      # a sixteenth of a cell
    for r in range(R):
        for _ in range(200):
            sc = rois[r, 1:5] * np.float32(scale)
            k = np.nonzero(on_half(sc))[0]
            if len(k):
                rois[r, 1 + k] += step
                continue
            sw, sh, ew, eh = rect(rois[r], scale)
            if edge_divergent(max(ew - sw + 1, 1), PW):
                rois[r, 3] += np.float32(1.0 / scale)        # one cell wider
                continue
            if edge_divergent(max(eh - sh + 1, 1), PH):
                rois[r, 4] += np.float32(1.0 / scale)
                continue
            break
        else:
            raise AssertionError("could not place roi %d of %s" % (r, name))
    return feat, np.ascontiguousarray(rois)


def check_clean(name, rois):
    _, shape, R, PH, PW, scale = _case(name)
    assert not on_half(rois[:, 1:5] * np.float32(scale)).any()
    for r in range(len(rois)):
        sw, sh, ew, eh = rect(rois[r], scale)
        assert not edge_divergent(max(ew - sw + 1, 1), PW) and not edge_divergent(max(eh - sh + 1, 1), PH)
