"""The operator boundary BY THE REFERENCE'S NAMES (extensions/__init__.py:1-3 and the module paths
tools/faster_rcnn_train_val.py / the model files import): after `scda_amd.dropin.install()` the packages `extensions`,
`models`, `functions`, `utils` resolve to the MI355X implementation; each callable is run forward AND backward through
its reference-shaped wrapper and compared with the CPU oracle (the raw C-ABI entry points are covered by
test_detection_ops_gpu.py -- this file covers what a user of the reference actually calls)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import native_ops as orc
from test_oracle_golden import rand_boxes, rand_rois

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def installed():
    import scda_amd.dropin as dropin
    dropin.install()


def test_extensions_nms_and_roipool(cuda):
    from extensions import nms, RoIPool
    rs = np.random.RandomState(5)
    b = rand_boxes(rs, 3000)
    b[1500:, :4] = b[:1500, :4] + rs.uniform(-3, 3, (1500, 4)).astype(np.float32)
    for dets in (torch.from_numpy(b), torch.from_numpy(b).to(cuda)):      # the reference hands over a CPU tensor (rpn_proposal.py:64)
        keep = nms(dets, 0.7)
        assert keep.dtype == torch.int64 and keep.device.type == "cpu" and keep.is_contiguous()
        assert np.array_equal(keep.numpy(), orc.nms(b, 0.7))
    with pytest.raises(ValueError):
        nms(torch.zeros(4, 4), 0.5)

    pool = RoIPool(7, 7, 1.0 / 16)
    assert isinstance(pool, torch.nn.Module) and (pool.pooled_height, pool.pooled_width) == (7, 7)
    feat = rs.randn(2, 24, 20, 34).astype(np.float32)
    rois = rand_rois(rs, 40, 2, 34 * 16, 20 * 16)
    x = torch.from_numpy(feat).to(cuda).requires_grad_()
    out = pool(x, torch.from_numpy(rois).to(cuda))
    eo, ea = orc.roi_pool_fwd(feat, rois, 7, 7, 1 / 16.)
    assert np.array_equal(out.detach().cpu().numpy(), eo)
    top = rs.randn(*eo.shape).astype(np.float32)
    out.backward(torch.from_numpy(top).to(cuda))
    assert np.array_equal(x.grad.cpu().numpy(), orc.roi_pool_bwd(top, ea, rois, feat.shape, 7, 7, 1 / 16.))
    with pytest.raises(AssertionError):                                    # functions/roi_pool.py:25-26, modules/roi_pool.py:13
        pool(x, torch.zeros(3, 4, device=cuda))
    with pytest.raises(AssertionError):
        pool(x.transpose(2, 3), torch.from_numpy(rois).to(cuda))


@pytest.mark.parametrize("kind", ["RoIAlign", "RoIAlignAvg", "RoIAlignMax"])
def test_roi_align_modules_fwd_bwd(cuda, kind):
    """modules/roi_align.py:6-44: RoIAlign = the function; Avg / Max = align to (h+1, w+1), then 2x2 stride-1 pooling"""
    import extensions._roi_align.modules.roi_align as M
    rs = np.random.RandomState(9)
    feat = rs.randn(2, 16, 25, 42).astype(np.float32)
    rois = rand_rois(rs, 30, 2, 42 * 16, 25 * 16)
    mod = getattr(M, kind)(7, 7, 1.0 / 16)
    assert (mod.aligned_height, mod.aligned_width, mod.spatial_scale) == (7, 7, 1.0 / 16)
    x = torch.from_numpy(feat).to(cuda).requires_grad_()
    out = mod(x, torch.from_numpy(rois).to(cuda))
    extra = 0 if kind == "RoIAlign" else 1
    ref_in = torch.from_numpy(orc.roi_align_fwd(feat, rois, 7 + extra, 7 + extra, 1 / 16.)).requires_grad_()
    ref = ref_in if kind == "RoIAlign" else (F.avg_pool2d if kind == "RoIAlignAvg" else F.max_pool2d)(ref_in, kernel_size=2, stride=1)
    assert tuple(out.shape) == (30, 16, 7, 7)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-6)
    top = torch.from_numpy(rs.randn(30, 16, 7, 7).astype(np.float32))
    out.backward(top.to(cuda))
    ref.backward(top)
    want = orc.roi_align_bwd(ref_in.grad.numpy(), rois, feat.shape, 7 + extra, 7 + extra, 1 / 16.)
    np.testing.assert_allclose(x.grad.cpu().numpy(), want, rtol=1e-4, atol=1e-5)    # backward accumulates with atomics
    with pytest.raises(NotImplementedError):                                          # functions/roi_align.py:30-31
        from extensions._roi_align.functions.roi_align import RoIAlignFunction
        RoIAlignFunction(7, 7, 1 / 16.)(torch.from_numpy(feat), torch.from_numpy(rois))


def test_focal_loss_functions_fwd_bwd(cuda):
    """focal_loss.py:7-142: callable(gamma, alpha, num_classes)(preds, targets, weight_pos) -> 1-element loss; backward =
    kernel gradient * grad_output"""
    from extensions._focal_loss.focal_loss import SigmoidFocalLossFunction, SoftmaxFocalLossFunction
    rs = np.random.RandomState(3)
    R, C = 777, 8
    x = (rs.randn(R, C) * 2).astype(np.float32)
    wp = torch.tensor([37.0])
    # sigmoid variant: targets in 0..C (0 = background), -1 ignored
    t = rs.randint(-1, C + 1, R).astype(np.int32)
    p = torch.from_numpy(x).to(cuda).requires_grad_()
    loss = SigmoidFocalLossFunction(2.0, 0.25, C)(p, torch.from_numpy(t).to(cuda), wp)
    assert tuple(loss.shape) == (1,)
    want = orc.focal_sigmoid_fwd(x, t, 37.0, 2.0, 0.25, C)
    np.testing.assert_allclose(float(loss), float(want.astype(np.float64).sum()), rtol=2e-6)
    (0.5 * loss).sum().backward()
    np.testing.assert_allclose(p.grad.cpu().numpy(), 0.5 * orc.focal_sigmoid_bwd(x, t, 37.0, 2.0, 0.25, C), atol=1e-6, rtol=1e-5)
    # softmax variant: targets in 0..C-1, -1 ignored
    t2 = rs.randint(-1, C, R).astype(np.int32)
    p2 = torch.from_numpy(x).to(cuda).requires_grad_()
    loss2 = SoftmaxFocalLossFunction(2.0, 0.25, C)(p2, torch.from_numpy(t2).to(cuda), wp)
    el, ep = orc.focal_softmax_fwd(x, t2, 37.0, 2.0, 0.25, C)
    np.testing.assert_allclose(float(loss2), float(el.astype(np.float64).sum()), rtol=2e-6)
    (2.0 * loss2).sum().backward()
    np.testing.assert_allclose(p2.grad.cpu().numpy(), 2.0 * orc.focal_softmax_bwd(x, t2, ep, 37.0, 2.0, 0.25, C), atol=2e-6, rtol=1e-5)
    with pytest.raises(AssertionError):
        SigmoidFocalLossFunction(2.0, 0.25, C + 1)(p, torch.from_numpy(t).to(cuda), wp)


def test_box_helpers(cuda, golden_dir):
    from extensions._bbox_helper.bbox_helper import overlap
    from extensions._cython_bbox import cython_bbox, cython_nms
    import utils.bbox_helper as bh
    rs = np.random.RandomState(11)
    b1, b2 = rand_boxes(rs, 300), rand_boxes(rs, 17, integer=True)
    o = overlap(b1, b2)
    assert isinstance(o, np.ndarray) and o.shape == (300, 17)
    assert np.array_equal(o, orc.iou_overlaps(b1[:, :4], b2[:, :4]))
    g = np.load(os.path.join(golden_dir, "bbox_overlaps.npz"))          # the reference's own Cython output
    for case in ("small", "anchors", "degenerate"):
        assert np.array_equal(cython_bbox.bbox_overlaps(g[case + "_boxes"], g[case + "_query"]), g[case + "_out"])
        assert np.array_equal(bh.bbox_iou_overlaps(g[case + "_boxes"], g[case + "_query"]), g[case + "_out"])
    # cython_nms.nms: scores in ANY order, ">=" threshold, returns np.where(kept)[0] (cython_nms.pyx:37-87) == the nms.c variant
    d = rand_boxes(rs, 400)
    d[200:, :4] = d[:200, :4] + rs.uniform(-2, 2, (200, 4)).astype(np.float32)
    d[:, 4] = rs.permutation(d[:, 4])
    keep = cython_nms.nms(d, np.float32(0.6))
    order = d[:, 4].argsort()[::-1]
    areas = (d[:, 2] - d[:, 0] + 1) * (d[:, 3] - d[:, 1] + 1)
    assert np.array_equal(keep, np.sort(orc.cpu_nms(d, order, areas, 0.6)))
    # soft_nms, method 0 (hard suppression above Nt) keeps the greedy-NMS set (cython_nms.pyx:95-203)
    ds = d[order]
    boxes, inds = cython_nms.soft_nms(ds, 0.5, 0.3, 0.001, 0)
    assert set(int(i) for i in inds) == set(int(i) for i in orc.nms(ds, 0.3)) and boxes.shape == (len(inds), 5)
