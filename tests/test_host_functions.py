"""Host-side box logic (scda_amd/dropin/functions, utils) against golden vectors produced by the reference's own
Python (tests/golden/make_golden.py).  CPU only: the IoU / NMS hooks are pointed at the C oracle; the same
assertions run with the HIP kernels in tests/test_host_functions_gpu.py."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import native_ops as orc
from scda_amd.dropin import backend

CFG = {
    "shared": {"anchor_scales": [2, 4, 8, 16, 32], "anchor_ratios": [0.5, 1, 2], "anchor_stride": 16,
               "bbox_normalize_stats_precomputed": True, "bbox_normalize_stds": [0.1, 0.1, 0.2, 0.2],
               "bbox_normalize_means": [0, 0, 0, 0], "num_classes": 9},
    "train_anchor_target_cfg": {"rpn_batch_size": 256, "nms_iou_thresh": 0.7, "positive_iou_thresh": 0.7,
                                "negative_iou_thresh": 0.3, "positive_percent": 0.5, "ignore_iou_thresh": 0.5},
    "train_rpn_proposal_cfg": {"nms_iou_thresh": 0.7, "pre_nms_top_n": 12000, "post_nms_top_n": 2000, "roi_min_size": 2},
    "train_proposal_target_cfg": {"batch_size": 512, "positive_iou_thresh": 0.5, "negative_iou_thresh_hi": 0.5,
                                  "negative_iou_thresh_lo": 0.0, "ignore_iou_thresh": 0.5, "positive_percent": 0.25,
                                  "append_gts": True},
    "test_rpn_proposal_cfg": {"nms_iou_thresh": 0.7, "pre_nms_top_n": 6000, "post_nms_top_n": 300, "roi_min_size": 2},
    "test_predict_bbox_cfg": {"nms_iou_thresh": 0.5, "score_thresh": 0.0, "top_n": 100},
}
for k in CFG:
    if k != "shared":
        CFG[k].update(CFG["shared"])


def synth_rpn_outputs(seed, A=15, fh=32, fw=64):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(1, fh, fw, A, 2, generator=g) * 2.0
    prob = torch.softmax(logits, -1)
    cls = prob.reshape(1, fh, fw, A * 2).permute(0, 3, 1, 2).contiguous()
    loc = (torch.randn(1, A * 4, fh, fw, generator=g) * 0.3).contiguous()
    return cls, loc


@pytest.fixture()
def cpu_backend():
    backend.use(bbox_overlaps=lambda b, q: orc.bbox_overlaps(b[:, :4], q[:, :4]),
                nms=lambda d, t: torch.from_numpy(orc.nms(d.numpy(), t)))
    yield
    backend.reset()


def check_l2(golden_dir, G, device=None):
    """device given: ground truth and RPN outputs are handed over as device tensors, which routes anchor labelling and proposal
    generation through the device-resident kernels (scda_amd/device_boxes.py); results are compared on the host either way"""
    from scda_amd.dropin.functions.anchor_target import compute_anchor_targets
    from scda_amd.dropin.functions.rpn_proposal import compute_rpn_proposals
    from scda_amd.dropin.functions.proposal_target import compute_proposal_targets
    from scda_amd.dropin.functions.mask import compute_cluster_targets
    g = np.load(os.path.join(golden_dir, f"l2_G{G}.npz"))
    seed = int(g["seed"])
    gts = torch.from_numpy(g["gts"]); info = torch.from_numpy(g["image_info"])
    if device is not None:
        gts_in = gts.to(device)
        gts_in._scda_host = g["gts"]
    else:
        gts_in = gts

    np.random.seed(seed)
    cls_t, loc_t, loc_m, norm = compute_anchor_targets((1, 60, 32, 64), CFG["train_anchor_target_cfg"], gts_in, info, None)
    if device is not None:
        assert cls_t.is_cuda and loc_t.is_cuda
        cls_t, loc_t, loc_m = cls_t.cpu(), loc_t.cpu(), loc_m.cpu()
    np.testing.assert_array_equal(cls_t.numpy().astype(np.int8), g["at_cls_targets"])
    assert norm == int(g["at_normalizer"])
    nz = np.nonzero(loc_m.numpy().reshape(-1))[0]
    np.testing.assert_array_equal(nz.astype(np.int32), g["at_loc_nz_index"])
    np.testing.assert_array_equal(loc_t.numpy().reshape(-1)[nz], g["at_loc_targets_nz"])
    assert cls_t.dtype == torch.int64 and loc_t.dtype == torch.float32 and tuple(loc_t.shape) == (1, 60, 32, 64)

    cls, loc = synth_rpn_outputs(seed)
    if device is not None:
        cls, loc = cls.to(device), loc.to(device)
    props = compute_rpn_proposals(cls, loc, CFG["train_rpn_proposal_cfg"], g["image_info"])
    assert not props.is_cuda and (device is None or torch.equal(props._scda_dev.cpu(), props))
    np.testing.assert_array_equal(props.numpy(), g["proposals"])
    props_test = compute_rpn_proposals(cls, loc, CFG["test_rpn_proposal_cfg"], g["image_info"])
    np.testing.assert_array_equal(props_test.numpy(), g["proposals_test"])

    np.random.seed(seed + 1)
    rois, labels, pt, pw = compute_proposal_targets(props, CFG["train_proposal_target_cfg"], gts_in, info, None)
    if device is not None:      # matched, compacted, gathered on the device (device_boxes.proposal_targets); the sampled rows' host copy rides along
        assert rois.is_cuda and labels.is_cuda and pt.is_cuda and labels.dtype == torch.int64
        np.testing.assert_array_equal(rois._scda_host, g["pt_rois"])
        rois, labels, pt, pw = rois.cpu(), labels.cpu(), pt.cpu(), pw.cpu()
    np.testing.assert_array_equal(rois.numpy(), g["pt_rois"])
    np.testing.assert_array_equal(labels.numpy().astype(np.int16), g["pt_labels"])
    nzp = np.nonzero(pw.numpy().reshape(-1))[0]
    np.testing.assert_array_equal(nzp.astype(np.int32), g["pt_loc_nz_index"])
    np.testing.assert_array_equal(pt.numpy().reshape(-1)[nzp], g["pt_loc_targets_nz"])
    assert rois.shape == (512, 5) and pt.shape == (512, 36)

    feats = torch.arange(512, dtype=torch.float32)[:, None].repeat(1, 8)
    np.random.seed(seed + 2)
    cf, centres = compute_cluster_targets(rois, feats, N_cluster=4, threshold=128)
    np.testing.assert_array_equal(cf.numpy()[:, :, 0].astype(np.int16), g["ct_index"])
    # sklearn reduces float32 partial sums per OpenMP thread: centres are reproducible to ~1e-4 px across thread counts
    np.testing.assert_allclose(centres, g["ct_centres"], rtol=0, atol=1e-3)
    assert not cf.requires_grad
    pg = props[0:512, :5].contiguous()
    np.random.seed(seed + 3)
    cf2, centres2 = compute_cluster_targets(pg, feats, N_cluster=4, threshold=128)
    np.testing.assert_array_equal(cf2.numpy()[:, :, 0].astype(np.int16), g["ct2_index"])
    np.testing.assert_allclose(centres2, g["ct2_centres"], rtol=0, atol=1e-3)


@pytest.mark.parametrize("G", [3, 12, 30])
def test_l2_functions_match_reference(golden_dir, cpu_backend, G):
    check_l2(golden_dir, G)


def check_predict(golden_dir):
    from scda_amd.dropin.functions.predict_bbox import compute_predicted_bboxes
    g = np.load(os.path.join(golden_dir, "predict_bbox.npz"))
    bb = compute_predicted_bboxes(torch.from_numpy(g["rois"]), torch.from_numpy(g["pred_cls"]), torch.from_numpy(g["pred_loc"]),
                                  g["image_info"], CFG["test_predict_bbox_cfg"])
    np.testing.assert_array_equal(bb.numpy(), g["bboxes"])


def test_predict_bbox_matches_reference(golden_dir, cpu_backend):
    check_predict(golden_dir)


def test_anchor_grid_matches_reference(golden_dir):
    from scda_amd.dropin.utils import anchor_helper
    g = np.load(os.path.join(golden_dir, "l2_G3.npz"))
    a = anchor_helper.get_anchors_over_plane(32, 64, [0.5, 1, 2], [2, 4, 8, 16, 32], 16)
    assert a.dtype == np.float64
    np.testing.assert_array_equal(a, g["anchors"])
    # the reference ignores anchor_ratios (utils/anchor_helper.py:10-11): so do we
    np.testing.assert_array_equal(anchor_helper.get_anchors_over_plane(32, 64, [7.0], [2, 4, 8, 16, 32], 16), a)


def test_crop_corners_match_reference(golden_dir):
    """a9: the 256 x 256 crop window around each cluster centre, pushed inside the 1024 x 512 image, against the output of the
    reference's own get_corner_from_center (tools/faster_rcnn_train_val.py:411-438) on 64 centres (tests/golden/make_golden.py:
    gen_corners) -- both clamp edges ((0, 0), (1023.9, 511.9)), centres on .5 (int() truncation) and the x2 == new_w / y2 == new_h
    branches that move the window's first corner"""
    from scda_amd.train_step import get_corner_from_center
    g = np.load(os.path.join(golden_dir, "corners.npz"))
    got = np.array(get_corner_from_center(g["centres"], 256, 1024, 512), dtype=np.int32)
    np.testing.assert_array_equal(got, g["corners"])
    # every window is exactly recon x recon and inside the image (what _crops asserts before slicing)
    assert np.all(got[:, 2] - got[:, 0] == 256) and np.all(got[:, 3] - got[:, 1] == 256)
    assert got[:, :2].min() >= 0 and got[:, 2].max() <= 1024 and got[:, 3].max() <= 512
    # the fixture exercises all four clamps
    assert (got[:, 0] == 0).any() and (got[:, 1] == 0).any() and (got[:, 2] == 1024).any() and (got[:, 3] == 512).any()
    # rectangular windows (the ResNet configuration's 128 x 256 reconstructions): the square case is the (s, s) special case
    np.testing.assert_array_equal(np.array(get_corner_from_center(g["centres"], (256, 256), 1024, 512), dtype=np.int32), g["corners"])


def test_proposal_targets_pad_by_resampling(cpu_backend):
    """fewer candidates than batch_size -> padded with replacement to exactly 512 (proposal_target.py:149-155)"""
    from scda_amd.dropin.functions.proposal_target import compute_proposal_targets
    rs = np.random.RandomState(1)
    props = np.zeros((40, 6), np.float32)
    props[:, 1:3] = rs.uniform(0, 400, (40, 2)); props[:, 3:5] = props[:, 1:3] + rs.uniform(20, 100, (40, 2))
    gts = torch.tensor([[[50, 50, 150, 150, 3], [300, 200, 420, 330, 7]]], dtype=torch.float32)
    np.random.seed(0)
    rois, labels, t, w = compute_proposal_targets(torch.from_numpy(props), CFG["train_proposal_target_cfg"], gts,
                                                  torch.tensor([[512, 1024, 1.0]]), None)
    assert rois.shape == (512, 5) and labels.shape == (512,) and t.shape == (512, 36)
    assert set(np.unique(labels.numpy())) <= {0, 3, 7}
    fg = labels.numpy() > 0
    assert (w.numpy()[fg].sum(1) == 4).all() and (w.numpy()[~fg].sum(1) == 0).all()


def test_empty_gt_image_is_skipped(cpu_backend):
    from scda_amd.dropin.functions.proposal_target import compute_proposal_targets
    props = torch.tensor([[0, 10, 10, 60, 60, .9], [1, 10, 10, 60, 60, .8]])
    gts = torch.zeros(2, 1, 5); gts[1, 0] = torch.tensor([12, 12, 58, 58, 2])
    np.random.seed(0)
    rois, labels, _, _ = compute_proposal_targets(props, CFG["train_proposal_target_cfg"], gts, torch.tensor([[512, 1024, 1.], [512, 1024, 1.]]))
    assert rois.shape[0] == 512 and (rois[:, 0] == 1).all()
