"""Box arithmetic on the host -- the contract of utils/bbox_helper.py (IoU, IoF, encode/decode, clip, recall).
The pairwise IoU goes to the MI355X (scda_bbox_overlaps_hip) instead of the reference's Cython loop."""
import warnings

import numpy as np

from scda_amd.dropin import backend


def bbox_iou_overlaps(b1, b2):
    """[n,>=4] x [m,>=4] -> fp32 [n,m]; no +1, zero unless the boxes truly intersect (utils/bbox_helper.py:8-9)."""
    return backend.bbox_overlaps(b1.astype(np.float32), b2.astype(np.float32))


def bbox_iof_overlaps(b1, b2):
    """intersection over the FIRST box's area (utils/bbox_helper.py:29-44)"""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    ix = np.minimum(b1[:, 2, None], b2[None, :, 2]) - np.maximum(b1[:, 0, None], b2[None, :, 0])
    iy = np.minimum(b1[:, 3, None], b2[None, :, 3]) - np.maximum(b1[:, 1, None], b2[None, :, 1])
    inter = np.maximum(ix, 0) * np.maximum(iy, 0)
    return inter / np.maximum(a1[:, None], 1)


def center_to_corner(boxes):
    half_w, half_h = boxes[:, 2] / 2., boxes[:, 3] / 2.
    return np.stack([boxes[:, 0] - half_w, boxes[:, 1] - half_h, boxes[:, 0] + half_w, boxes[:, 1] + half_h], axis=1)


def corner_to_center(boxes):
    return np.stack([(boxes[:, 0] + boxes[:, 2]) / 2., (boxes[:, 1] + boxes[:, 3]) / 2.,
                     boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], axis=1)


def compute_loc_targets(raw_bboxes, gt_bboxes):
    """(dx, dy, log dw, log dh) of gt relative to raw (utils/bbox_helper.py:70-86)"""
    r, g = corner_to_center(raw_bboxes), corner_to_center(gt_bboxes)
    assert np.all(r[:, 2] > 0) and np.all(r[:, 3] > 0)
    return np.stack([(g[:, 0] - r[:, 0]) / r[:, 2], (g[:, 1] - r[:, 1]) / r[:, 3],
                     np.log(g[:, 2] / r[:, 2]), np.log(g[:, 3] / r[:, 3])], axis=1)


def compute_loc_bboxes(raw_bboxes, deltas):
    """inverse of compute_loc_targets (utils/bbox_helper.py:88-103)"""
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        r = corner_to_center(raw_bboxes)
        ctr = np.stack([deltas[:, 0] * r[:, 2] + r[:, 0], deltas[:, 1] * r[:, 3] + r[:, 1],
                        np.exp(deltas[:, 2]) * r[:, 2], np.exp(deltas[:, 3]) * r[:, 3]], axis=1)
        return center_to_corner(ctr)


def clip_bbox(bbox, img_size):
    h, w = img_size[:2]
    for col, hi in ((0, w - 1), (1, h - 1), (2, w - 1), (3, h - 1)):
        bbox[:, col] = np.clip(bbox[:, col], 0, hi)
    return bbox


def compute_recall(box_pred, box_gt):
    n_gt = box_gt.shape[0]
    if box_pred.size == 0 or n_gt == 0:
        return 0, n_gt
    best = bbox_iou_overlaps(box_gt, box_pred).max(axis=1)
    return int((best > 0.5).sum()), n_gt
