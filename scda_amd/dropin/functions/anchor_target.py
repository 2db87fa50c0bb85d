"""RPN anchor labelling -- the contract of functions/anchor_target.py:16-116.

Host-side numpy (the north star keeps host logic in Python); the anchors-vs-gt IoU matrix is computed on the
MI355X.  The numpy global RNG is consumed in exactly the reference's order (one np.random.choice for surplus
positives, one for surplus negatives) so that seeded runs select the same anchors."""
import numpy as np
import torch

from scda_amd.dropin import backend
from scda_amd.dropin.utils import anchor_helper, bbox_helper


def _np(x):
    return backend.host_array(x)


def _to_dev(a, like):
    t = torch.from_numpy(a)
    if torch.is_tensor(like) and like.is_cuda:
        from scda_amd import native
        return native.upload(t, like.device)
    return t.to(like.device) if torch.is_tensor(like) else t


def compute_anchor_targets(feature_size, cfg, ground_truth_bboxes, image_info, ignore_regions=None):
    """-> cls_targets int64 [B,A,h,w] (1 fg / 0 bg / -1 ignore), loc_targets, loc_masks fp32 [B,4A,h,w], normaliser"""
    if (ignore_regions is None and feature_size[0] == 1 and torch.is_tensor(ground_truth_bboxes) and ground_truth_bboxes.is_cuda
            and ground_truth_bboxes.dtype == torch.float32):
        from scda_amd import device_boxes
        if device_boxes.enabled():     # IoU, labels, sub-sampling and the target maps stay on the MI355X (box_ops.hip)
            return device_boxes.anchor_targets(feature_size, cfg, ground_truth_bboxes)
    dev_like = ground_truth_bboxes
    gts, image_info, ignore_regions = _np(ground_truth_bboxes), _np(image_info), _np(ignore_regions)
    B, A4, fh, fw = feature_size
    A = A4 // 4
    assert A * 4 == A4
    anchors = anchor_helper.get_anchors_over_plane(fh, fw, cfg['anchor_ratios'], cfg['anchor_scales'], cfg['anchor_stride'])
    KA = anchors.shape[0]

    iou = np.stack([bbox_helper.bbox_iou_overlaps(anchors, gts[b]) for b in range(B)], axis=0)  # [B,KA,G]
    best_gt = iou.argmax(axis=2)
    best_iou = np.take_along_axis(iou, best_gt[:, :, None], axis=2)[:, :, 0]   # == iou.max(axis=2), one pass less
    per_gt_best = np.stack([iou[b].max(axis=0) for b in range(B)], axis=0)      # [B,G] (contiguous column reduce)
    per_gt_best[per_gt_best < 0.1] = -1  # a gt nobody overlaps by >= 0.1 claims no anchor
    gb, gka, gg = np.where(iou == per_gt_best[:, None, :])
    best_gt[gb, gka] = gg

    labels = np.full((B, KA), -1, dtype=np.int64)
    labels[best_iou < cfg['negative_iou_thresh']] = 0
    if ignore_regions is not None:
        iof = np.stack([bbox_helper.bbox_iof_overlaps(anchors, ignore_regions[b]) for b in range(B)], axis=0)
        labels[iof.max(axis=2) > cfg['ignore_iou_thresh']] = -1
    labels[gb, gka] = 1
    labels[best_iou > cfg['positive_iou_thresh']] = 1

    # subsample to rpn_batch_size per image, at most positive_percent of it foreground
    budget = cfg['rpn_batch_size'] * B
    max_pos = int(cfg['positive_percent'] * budget)
    pb, pk = np.where(labels > 0)
    n_pos = len(pb)
    if n_pos > max_pos:
        drop = np.random.choice(n_pos, size=n_pos - max_pos, replace=False)
        labels[pb[drop], pk[drop]] = -1
        n_pos = max_pos
    max_neg = budget - n_pos
    nb, nk = np.where(labels == 0)
    if len(nb) > max_neg:
        drop = np.random.choice(len(nb), size=len(nb) - max_neg, replace=False)
        labels[nb[drop], nk[drop]] = -1

    pb, pk = np.where(labels > 0)
    matched = gts[pb, best_gt[pb, pk]]
    deltas = bbox_helper.compute_loc_targets(anchors[pk, :], matched)
    loc_t = np.zeros((B, KA, 4), dtype=np.float32)
    loc_m = np.zeros((B, KA, 4), dtype=np.float32)
    loc_t[pb, pk, :] = deltas
    loc_m[pb, pk, :] = 1.

    def to_map(a, ch):  # [B,KA,c] -> [B, A*c, fh, fw] on the gt's device
        return _to_dev(a, dev_like).view(B, fh, fw, ch).permute(0, 3, 1, 2).contiguous()

    cls_targets = to_map(labels, A).long()
    loc_targets = to_map(loc_t, A * 4).float()
    loc_masks = to_map(loc_m, A * 4).float()
    normalizer = max(1, int((labels >= 0).sum()))
    return cls_targets, loc_targets, loc_masks, normalizer
