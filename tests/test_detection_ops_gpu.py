"""Parity of the HIP detection operators (through the C ABI, scda_amd.native) with the CPU oracle.
Integer / index results must be bit-exact; focal loss within 1e-6 abs (device expf/logf/powf)."""
import numpy as np
import pytest
import torch

from oracle import native_ops as orc
from test_oracle_golden import rand_boxes, rand_rois

pytestmark = pytest.mark.gpu


def dev(a, cuda, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(cuda).contiguous()


def clustered(rs, n, integer=False):
    b = rand_boxes(rs, n, integer=integer)
    if n > 10:
        b[n // 2:, :4] = b[: n - n // 2, :4] + rs.uniform(-3, 3, (n - n // 2, 4)).astype(np.float32)
    return b


@pytest.mark.parametrize("n,thresh,integer", [(0, .7, False), (1, .7, False), (63, .7, False), (64, .7, True), (65, .5, False),
                                               (300, .5, False), (2000, .7, True), (6000, .7, False), (12000, .7, False)])
def test_nms_keep_bit_exact(cuda, n, thresh, integer):
    from scda_amd import native
    rs = np.random.RandomState(n + 7)
    boxes = clustered(rs, n, integer)
    ref = orc.nms(boxes, thresh)
    keep, num = native.nms(dev(boxes, cuda) if n else torch.zeros(0, 5, device=cuda), thresh)
    k = int(num.item())
    assert k == len(ref)
    np.testing.assert_array_equal(keep[:k].cpu().numpy(), ref)


def test_nms_keep_equals_reference_cython_nms(cuda, golden_dir):
    """scda_nms_hip against keep lists produced by the REFERENCE's cython_nms.pyx compiled unmodified
    (tests/golden/nms_ref.npz, make_golden_nms.py): bit-exact at 300 ... 12000 boxes, RPN-like / clustered / integer boxes.
    Also through the operator boundary (`from extensions import nms`) and through the validity-flag sweep of the
    device-resident proposal path with every box valid."""
    from scda_amd import dropin, native
    from test_oracle_golden import _nms_ref
    dropin.install()
    from extensions import nms as ext_nms
    for name, dets, thresh, ref in _nms_ref(golden_dir):
        d = dev(dets, cuda)
        keep, num = native.nms(d, thresh)
        k = int(num.item())
        np.testing.assert_array_equal(keep[:k].cpu().numpy(), ref, err_msg=name)
        np.testing.assert_array_equal(ext_nms(d, thresh).cpu().numpy(), ref, err_msg=name + " (extensions.nms)")
        keep, num = native.nms(d, thresh, max_keep=2000)
        np.testing.assert_array_equal(keep[:int(num.item())].cpu().numpy(), ref[:2000], err_msg=name + " (max_keep)")


@pytest.mark.parametrize("n,max_keep", [(500, 10), (3000, 300), (12000, 2000), (700, 100000)])
def test_nms_max_keep_equals_truncation(cuda, n, max_keep):
    from scda_amd import native
    rs = np.random.RandomState(n)
    boxes = clustered(rs, n)
    ref = orc.nms(boxes, 0.7)[:max_keep]
    keep, num = native.nms(dev(boxes, cuda), 0.7, max_keep=max_keep)
    k = int(num.item())
    assert k == len(ref)
    np.testing.assert_array_equal(keep[:k].cpu().numpy(), ref)


def test_nms_mask_upper_triangle_bit_exact(cuda):
    from scda_amd import native
    rs = np.random.RandomState(3)
    boxes = clustered(rs, 1000)
    m = native.nms_mask(dev(boxes, cuda), 0.6).cpu().numpy().view(np.uint64)
    ref = orc.nms_mask(boxes, 0.6)
    cb = ref.shape[1]
    rows = np.arange(1000) // 64
    upper = np.arange(cb)[None, :] >= rows[:, None]
    np.testing.assert_array_equal(m[upper], ref[upper])


def test_nms_segments_equal_per_list_nms(cuda):
    """scda_nms_segments_hip (the per-class lists of functions/predict_bbox.py:29-55 in one mask + one sweep launch): every list's
    keep indices equal a separate scda_nms_hip call's and the oracle's, for empty / one-box / chunk-boundary / 300-box lists"""
    from scda_amd import native
    from scda_amd.dropin import backend
    rs = np.random.RandomState(12)
    lens = [0, 1, 63, 64, 65, 300, 128, 0, 300, 7]
    lists = []
    for n in lens:
        b = rand_boxes(rs, n) if n else np.zeros((0, 5), np.float32)
        if n:
            b = b[np.argsort(-b[:, 4], kind="stable")]
        lists.append(b.astype(np.float32))
    got = backend.nms_segments(lists, 0.5)
    assert len(got) == len(lists)
    for b, k in zip(lists, got):
        want = orc.nms(b, 0.5) if b.shape[0] else np.zeros(0, np.int64)
        np.testing.assert_array_equal(np.asarray(k), want)
        if b.shape[0]:
            keep, num = native.nms(dev(b, cuda), 0.5)
            np.testing.assert_array_equal(keep[: int(num)].cpu().numpy(), want)


def test_nms_rejects_cpu_tensor(cuda):
    from scda_amd import native
    with pytest.raises(native.ScdaNativeError):
        native.nms(torch.zeros(3, 5), 0.5)


# (1, 6, 80, 80): 6400 pixels per plane = 83 KB of dynamic LDS in the ordered-scatter backward (the 64 - 96 KB window that needs
# hipFuncAttributeMaxDynamicSharedMemorySize); (1, 3, 100, 120): 12000 pixels, beyond it -> the gather kernel
@pytest.mark.parametrize("shape,R", [((1, 8, 32, 64), 64), ((2, 5, 16, 24), 33), ((1, 512, 32, 64), 512), ((1, 6, 80, 80), 96),
                                     ((1, 3, 100, 120), 40)])
def test_roi_pool_fwd_bwd_bit_exact(cuda, shape, R):
    from scda_amd import native
    rs = np.random.RandomState(R)
    B, C, H, W = shape
    feat = rs.randn(*shape).astype(np.float32)
    feat[0, 0, 1:5, 2:9] = 2.5  # ties
    rois = rand_rois(rs, R, B=B, W=W * 16, H=H * 16)
    rois[-1] = [0, -50, -50, W * 16 + 80, H * 16 + 80]
    # tiny RoIs: all 49 bins collapse onto 1-4 feature pixels (the ordered-scatter backward must serialise them in bin order)
    rois[-2] = [0, 37, 21, 40, 24]
    rois[-3] = [B - 1, 100, 60, 119, 70]
    rois[-4] = rois[-2]
    eo, ea = orc.roi_pool_fwd(feat, rois, 7, 7, 1 / 16.)
    out, arg = native.roi_pool_fwd(dev(feat, cuda), dev(rois, cuda), 7, 7, 1 / 16.)
    np.testing.assert_array_equal(out.cpu().numpy(), eo)
    np.testing.assert_array_equal(arg.cpu().numpy(), ea)
    top = rs.randn(*eo.shape).astype(np.float32)
    g = native.roi_pool_bwd(dev(top, cuda), arg, dev(rois, cuda), shape, 7, 7, 1 / 16.).cpu().numpy()
    # every channel, bit for bit.  The scatter form of the oracle sums each input element's contributors in the gather kernel's
    # (roi, ph, pw) order (proved bit-identical to the O(B*C*H*W*R) gather restatement in test_oracle_golden.py) at 1/1000 the cost.
    np.testing.assert_array_equal(g, orc.roi_pool_bwd_scatter(top, ea, shape, 7, 7))
    if C * R <= 4096:
        np.testing.assert_array_equal(g, orc.roi_pool_bwd(top, ea, rois, shape, 7, 7, 1 / 16.))


@pytest.mark.parametrize("step", [2, 3, 5])
def test_roi_pool_bwd_on_shared_maxima(cuda, step):
    """the ordered-scatter backward where bins share their maxima all over the place: a feature map with one dominant pixel every
    `step` rows and columns, RoIs from 1 to 20 feature pixels a side (up to 28 bins of a RoI on one pixel) -- bit-identical to the
    oracle's (roi, ph, pw)-ordered sum, on planes the scatter kernel takes and on one that is too large for its LDS (gather kernel)"""
    from scda_amd import native
    rs = np.random.RandomState(100 + step)
    for shape, R in (((1, 16, 32, 64), 300), ((2, 4, 24, 40), 200), ((1, 2, 100, 120), 150)):
        B, C, H, W = shape
        feat = (0.01 * rs.randn(*shape)).astype(np.float32)
        feat[:, :, ::step, ::step] += 10.0 + rs.rand(B, C, (H + step - 1) // step, (W + step - 1) // step).astype(np.float32)
        rois = np.zeros((R, 5), np.float32)
        rois[:, 0] = rs.randint(0, B, R)
        wd = rs.randint(1, 21, R) * 16.0; ht = rs.randint(1, 21, R) * 16.0
        rois[:, 1] = rs.randint(-2, W - 1, R) * 16.0 + rs.randint(0, 16, R); rois[:, 2] = rs.randint(-2, H - 1, R) * 16.0 + rs.randint(0, 16, R)
        rois[:, 3] = rois[:, 1] + wd - 1; rois[:, 4] = rois[:, 2] + ht - 1
        eo, ea = orc.roi_pool_fwd(feat, rois, 7, 7, 1 / 16.)
        shared = sum(len(a) - len(np.unique(a)) for a in (ea[r, c].ravel()[ea[r, c].ravel() >= 0] for r in range(R) for c in range(C)))
        assert shared > R * C, shared          # the case under test occurs: more than one shared maximum per (RoI, channel) on average
        top = rs.randn(*eo.shape).astype(np.float32)
        g = native.roi_pool_bwd(dev(top, cuda), dev(ea, cuda), dev(rois, cuda), shape, 7, 7, 1 / 16.).cpu().numpy()
        np.testing.assert_array_equal(g, orc.roi_pool_bwd_scatter(top, ea, shape, 7, 7))


def test_roi_pool_equals_reference_roi_pool_py(cuda, golden_dir):
    """scda_roi_pool_fwd_hip against outputs of the REFERENCE's own roi_pool_py.py (tests/golden/roi_pool_ref.npz), bit for bit, at
    [1,512,32,64] x 512 RoIs and two ragged shapes; the argmax contract (first maximum in scan order, -1 for empty bins) derived from
    those values; the backward (not part of roi_pool_py.py) against the scatter oracle on the same RoIs."""
    from scda_amd import native
    from test_oracle_golden import _roipool_ref, check_argmax_first_max
    for name, feat, rois, PH, PW, scale, check in _roipool_ref(golden_dir):
        out, arg = native.roi_pool_fwd(dev(feat, cuda), dev(rois, cuda), PH, PW, scale)
        o, a = out.cpu().numpy(), arg.cpu().numpy()
        check(o)
        check_argmax_first_max(feat, rois, o, a, PH, PW, scale, every=16 if feat.shape[1] > 64 else 1)
        top = np.random.RandomState(3).standard_normal(o.shape).astype(np.float32)
        g = native.roi_pool_bwd(dev(top, cuda), arg, dev(rois, cuda), feat.shape, PH, PW, scale).cpu().numpy()
        np.testing.assert_array_equal(g, orc.roi_pool_bwd_scatter(top, a, feat.shape, PH, PW), err_msg=name)


def test_roi_pool_empty_rois(cuda):
    from scda_amd import native
    feat = torch.randn(1, 4, 8, 8, device=cuda)
    out, arg = native.roi_pool_fwd(feat, torch.zeros(0, 5, device=cuda), 7, 7, 1 / 16.)
    assert out.shape == (0, 4, 7, 7)
    g = native.roi_pool_bwd(torch.zeros(0, 4, 7, 7, device=cuda), arg, torch.zeros(0, 5, device=cuda), (1, 4, 8, 8), 7, 7, 1 / 16.)
    assert float(g.abs().sum()) == 0.0


def test_roi_align_fwd_bwd(cuda):
    from scda_amd import native
    rs = np.random.RandomState(21)
    shape = (2, 16, 50, 84)
    feat = rs.randn(*shape).astype(np.float32)
    rois = rand_rois(rs, 64, B=2, W=84 * 16, H=50 * 16)
    rois[0] = [0, -30, -30, 100, 100]
    eo = orc.roi_align_fwd(feat, rois, 8, 8, 1 / 16.)
    out = native.roi_align_fwd(dev(feat, cuda), dev(rois, cuda), 8, 8, 1 / 16.)
    np.testing.assert_array_equal(out.cpu().numpy(), eo)  # same roundings -> bit exact
    top = rs.randn(*eo.shape).astype(np.float32)
    eg = orc.roi_align_bwd(top, rois, shape, 8, 8, 1 / 16.)
    g = native.roi_align_bwd(dev(top, cuda), dev(rois, cuda), shape, 8, 8, 1 / 16.)
    np.testing.assert_allclose(g.cpu().numpy(), eg, rtol=1e-5, atol=1e-5)  # atomics: order differs


def _roi_align_third_statement(feat, rois, AH, AW, scale, top=None):
    """A THIRD, independent statement of extensions/_roi_align/src/roi_align_kernel.cu:15-70 (forward) and :94-143 (backward), written
    from the kernel text by flat-index arithmetic only -- it shares no helper with oracle/scda_oracle.c or tests/np_restate.py.  Sample
    geometry in float32 exactly as the kernel's float variables hold it (the `1.` literals make the divisions double, the results are
    stored to float), interpolation in float64.  -> out [R, C, AH, AW] (float64), and with `top` the gradient [B, C, H, W]."""
    B, C, H, W = feat.shape
    R = rois.shape[0]
    f32 = np.float32
    r = rois.astype(f32)
    sc = f32(scale)
    x1, y1, x2, y2 = (r[:, 1] * sc).astype(f32), (r[:, 2] * sc).astype(f32), (r[:, 3] * sc).astype(f32), (r[:, 4] * sc).astype(f32)
    rw = np.maximum((x2 - x1).astype(np.float64) + 1.0, 0.0).astype(f32)          # fmaxf(roi_end_w - roi_start_w + 1., 0.)
    rh = np.maximum((y2 - y1).astype(np.float64) + 1.0, 0.0).astype(f32)
    bh = (rh.astype(np.float64) / (AH - 1.0)).astype(f32)                          # roi_height / (aligned_height - 1.)
    bw = (rw.astype(np.float64) / (AW - 1.0)).astype(f32)
    ph = np.arange(AH, dtype=f32)[None, :]
    pw = np.arange(AW, dtype=f32)[None, :]
    h = (ph * bh[:, None] + y1[:, None]).astype(f32)                               # [R, AH]   (float)(ph) * bin_size_h + roi_start_h
    w = (pw * bw[:, None] + x1[:, None]).astype(f32)                               # [R, AW]
    hs = np.minimum(np.floor(h), f32(H - 2)).astype(np.int64)                      # fminf(floor(h), height - 2)
    ws = np.minimum(np.floor(w), f32(W - 2)).astype(np.int64)
    inside = ~((h < 0) | (h >= H))[:, :, None] & ~((w < 0) | (w >= W))[:, None, :] # [R, AH, AW]
    hr = (h - hs.astype(f32)).astype(f32).astype(np.float64)[:, :, None]           # h_ratio (float), used in double arithmetic
    wr = (w - ws.astype(f32)).astype(f32).astype(np.float64)[:, None, :]
    img = r[:, 0].astype(np.int64)
    base = img[:, None, None] * (C * H * W) + np.where(inside, hs[:, :, None] * W + ws[:, None, :], 0)   # channel 0's up-left corner
    flat = feat.astype(np.float64).reshape(-1)
    coef = [(1.0 - hr) * (1.0 - wr), (1.0 - hr) * wr, hr * (1.0 - wr), hr * wr]
    offs = [0, 1, W, W + 1]
    chan = (np.arange(C, dtype=np.int64) * (H * W))[None, :, None, None]
    idx0 = base[:, None, :, :] + chan                                               # [R, C, AH, AW]
    out = np.zeros((R, C, AH, AW))
    for cf, o in zip(coef, offs):
        out += flat[idx0 + o] * cf[:, None, :, :]
    out *= inside[:, None, :, :]
    if top is None:
        return out
    grad = np.zeros(B * C * H * W)
    t = top.astype(np.float64) * inside[:, None, :, :]
    for cf, o in zip(coef, offs):
        np.add.at(grad, (idx0 + o).reshape(-1), (t * cf[:, None, :, :]).reshape(-1))
    return out, grad.reshape(B, C, H, W)


def test_roi_align_unpinned_third_independent_statement(cuda):
    """RoIAlign's oracle cannot be pinned against a build or run of the reference (its only source is a .cu file); besides the C
    restatement (bit-exact above) and tests/np_restate.py this is a third statement, at the shape BASELINE configs[3] runs: features
    [1, 1024, 50, 84], 512 RoIs, 8 x 8 samples (channels in four chunks of 256 to bound the float64 temporaries)."""
    from scda_amd import native
    rs = np.random.RandomState(77)
    C, H, W, R = 1024, 50, 84, 512
    feat = rs.randn(1, C, H, W).astype(np.float32)
    rois = rand_rois(rs, R, B=1, W=W * 16, H=H * 16)
    rois[0] = [0, -30, -30, 100, 100]                       # samples left of / above the map: zeros, no gradient
    rois[1] = [0, W * 16 - 40, H * 16 - 40, W * 16 + 60, H * 16 + 60]      # ... right of / below it; the hstart = height - 2 clamp
    top = rs.randn(R, C, 8, 8).astype(np.float32)
    out = native.roi_align_fwd(dev(feat, cuda), dev(rois, cuda), 8, 8, 1 / 16.).cpu().numpy()
    g = native.roi_align_bwd(dev(top, cuda), dev(rois, cuda), feat.shape, 8, 8, 1 / 16.).cpu().numpy()
    for c0 in range(0, C, 256):
        want, wg = _roi_align_third_statement(feat[:, c0:c0 + 256], rois, 8, 8, 1 / 16., top[:, c0:c0 + 256])
        np.testing.assert_allclose(out[:, c0:c0 + 256], want, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(g[:, c0:c0 + 256], wg, rtol=1e-4, atol=1e-4 * float(np.abs(wg).max()))
    assert float(np.abs(out[0]).max()) > 0 and (out[0, :, 0, 0] == 0).all()      # RoI 0's first sample lies outside


@pytest.mark.parametrize("shape,R,a", [((1, 24, 50, 84), 512, 8), ((2, 5, 13, 9), 33, 7), ((1, 3, 200, 336), 40, 8)])
def test_roi_align_bwd_plane_and_atomic_forms_agree(cuda, shape, R, a, monkeypatch):
    """the LDS-resident plane kernel (planes <= 64 KB) and the global-atomic kernel (larger planes, SCDA_ROI_ALIGN_ATOMIC=1)
    against the oracle and against each other, at the ResNet-50 C4 head's geometry (512 RoIs on a 50 x 84 map)"""
    from scda_amd import native
    rs = np.random.RandomState(R)
    rois = rand_rois(rs, R, B=shape[0], W=shape[3] * 16, H=shape[2] * 16)
    top = rs.randn(R, shape[1], a, a).astype(np.float32)
    want = orc.roi_align_bwd(top, rois, shape, a, a, 1 / 16.)
    got = native.roi_align_bwd(dev(top, cuda), dev(rois, cuda), shape, a, a, 1 / 16.).cpu().numpy()
    monkeypatch.setenv("SCDA_ROI_ALIGN_ATOMIC", "1")
    got_atomic = native.roi_align_bwd(dev(top, cuda), dev(rois, cuda), shape, a, a, 1 / 16.).cpu().numpy()
    scale = np.abs(want).max()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5 * scale)
    np.testing.assert_allclose(got_atomic, want, rtol=1e-5, atol=1e-5 * scale)


def test_focal_sigmoid(cuda):
    from scda_amd import native
    rs = np.random.RandomState(22)
    rows, C = 30720, 8
    x = (rs.randn(rows, C) * 3).astype(np.float32)
    t = rs.randint(-1, C + 1, rows).astype(np.int32)
    for wp in (0.0, 97.0):
        l = native.focal_sigmoid_fwd(dev(x, cuda), dev(t, cuda), wp, 2.0, 0.25, C)
        np.testing.assert_allclose(l.cpu().numpy(), orc.focal_sigmoid_fwd(x, t, wp, 2.0, 0.25, C), atol=1e-6, rtol=1e-5)
        g = native.focal_sigmoid_bwd(dev(x, cuda), dev(t, cuda), wp, 2.0, 0.25, C)
        np.testing.assert_allclose(g.cpu().numpy(), orc.focal_sigmoid_bwd(x, t, wp, 2.0, 0.25, C), atol=1e-6, rtol=1e-5)


def test_focal_softmax(cuda):
    from scda_amd import native
    rs = np.random.RandomState(23)
    rows, C = 30720, 9
    x = (rs.randn(rows, C) * 3).astype(np.float32)
    t = rs.randint(-1, C, rows).astype(np.int32)
    l, p = native.focal_softmax_fwd(dev(x, cuda), dev(t, cuda), 50.0, 2.0, 0.25, C)
    el, ep = orc.focal_softmax_fwd(x, t, 50.0, 2.0, 0.25, C)
    np.testing.assert_allclose(p.cpu().numpy(), ep, atol=1e-6)
    np.testing.assert_allclose(l.cpu().numpy(), el, atol=1e-6, rtol=1e-5)
    g = native.focal_softmax_bwd(dev(x, cuda), dev(t, cuda), p, 50.0, 2.0, 0.25, C)
    np.testing.assert_allclose(g.cpu().numpy(), orc.focal_softmax_bwd(x, t, ep, 50.0, 2.0, 0.25, C), atol=1e-6, rtol=1e-5)


def test_box_overlaps_bit_exact(cuda, golden_dir):
    import os
    from scda_amd import native
    g = np.load(os.path.join(golden_dir, "bbox_overlaps.npz"))
    for case in ("small", "anchors", "degenerate"):
        out = native.bbox_overlaps(dev(g[case + "_boxes"], cuda), dev(g[case + "_query"], cuda))
        np.testing.assert_array_equal(out.cpu().numpy(), g[case + "_out"], err_msg=case)  # vs the reference's Cython
    rs = np.random.RandomState(24)
    a = rand_boxes(rs, 30720)[:, :4]; b = rand_boxes(rs, 30)[:, :4]
    np.testing.assert_array_equal(native.iou_overlaps(dev(a, cuda), dev(b, cuda)).cpu().numpy(), orc.iou_overlaps(a, b))
    np.testing.assert_array_equal(native.bbox_overlaps(dev(a, cuda), dev(b, cuda)).cpu().numpy(), orc.bbox_overlaps(a, b))
