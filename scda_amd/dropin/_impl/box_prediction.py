"""Detections from RCNN outputs at test time (the contract of the reference's functions/predict_bbox.py:13-66).

Pipeline per (class 1..C-1, image): decode the class's box deltas against the RoIs (de-normalised by the configured
stds/means), clip to the image, drop scores <= score_thresh, order by score, NMS on the device, tag with image and class;
finally keep the top_n detections of each image over all classes.  Row layout: (image, x1, y1, x2, y2, score, class).
Result order (class-major, image-minor, then the per-image top_n cut) is the reference's: ties in the final argsort resolve
the same way."""
import numpy as np
import torch

from scda_amd.dropin import backend
from scda_amd.dropin.utils import bbox_helper


def _candidates_of(scores, boxes, image_hw, cfg):
    """one (image, class) pair -> its boxes [k,5] clipped, thresholded and ordered by score (descending), or None"""
    boxes[:, :4] = bbox_helper.clip_bbox(boxes[:, :4], image_hw)
    if cfg['score_thresh'] > 0:
        above = np.where(scores > cfg['score_thresh'])[0]
        scores, boxes = scores[above], boxes[above]
    if scores.size == 0:
        return None
    return boxes[scores.argsort()[::-1], :]


def compute_predicted_bboxes(rois, pred_cls, pred_loc, image_info, cfg):
    out_device = rois.device if torch.is_tensor(rois) else torch.device('cpu')
    rois, pred_cls, pred_loc, image_info = (backend.host_array(t) for t in (rois, pred_cls, pred_loc, image_info))
    n_rois, n_cls = pred_cls.shape[0:2]
    assert n_rois == rois.shape[0]
    n_img = int(max(rois[:, 0].astype(np.int32)) + 1)
    members = [np.where(rois[:, 0] == b)[0] for b in range(n_img)]
    stds, means = np.array(cfg['bbox_normalize_stds'])[None, :], np.array(cfg['bbox_normalize_means'])[None, :]

    # every (class, image) list first, then ONE batched NMS over all of them (one upload, a mask and a sweep launch with one
    # workgroup per list, one download) instead of (classes - 1) x images round trips
    lists, tags = [], []
    for cls in range(1, n_cls):
        scores = pred_cls[:, cls].squeeze()
        deltas = pred_loc[:, 4 * cls:4 * cls + 4].squeeze()
        if cfg['bbox_normalize_stats_precomputed']:
            deltas = deltas * stds + means
        scored = np.hstack([bbox_helper.compute_loc_bboxes(rois[:, 1:5], deltas), scores[:, None]])
        for b, idx in enumerate(members):
            cand = _candidates_of(scores[idx], scored[idx], image_info[b], cfg)
            if cand is not None:
                lists.append(cand)
                tags.append((b, cls))
    rows = []
    for cand, keep, (b, cls) in zip(lists, backend.nms_segments(lists, cfg['nms_iou_thresh']), tags):
        kept = cand[np.asarray(keep, dtype=np.int64)]
        n = kept.shape[0]
        rows.append(np.hstack([np.full((n, 1), b), kept, np.full((n, 1), cls)]))
    rows = np.vstack(rows)
    if cfg['top_n'] > 0:
        best = []
        for b in range(n_img):
            of_b = rows[rows[:, 0] == b]
            best.append(of_b[of_b[:, -2].argsort()[::-1][:cfg['top_n']]])
        rows = np.vstack(best)
    return torch.from_numpy(rows).float().to(out_device)
