"""Test-time box prediction -- the contract of functions/predict_bbox.py:13-66: per class decode (x stds), clip,
score-sort, NMS, then top_n per image over all classes.  Output rows: (batch, x1, y1, x2, y2, score, class)."""
import numpy as np
import torch

from scda_amd.dropin import backend
from scda_amd.dropin.utils import bbox_helper


def _np(x):
    if x is None:
        return None
    return x.detach().cpu().numpy() if torch.is_tensor(x) else x


def compute_predicted_bboxes(rois, pred_cls, pred_loc, image_info, cfg):
    dev = rois.device if torch.is_tensor(rois) else torch.device('cpu')
    rois, pred_cls, pred_loc = map(_np, (rois, pred_cls, pred_loc))
    image_info = _np(image_info)
    N, C = pred_cls.shape[0:2]
    B = int(max(rois[:, 0].astype(np.int32)) + 1)
    assert N == rois.shape[0]
    found = []
    for c in range(1, C):
        score = pred_cls[:, c].squeeze()
        delta = pred_loc[:, c * 4:c * 4 + 4].squeeze()
        if cfg['bbox_normalize_stats_precomputed']:
            delta = delta * np.array(cfg['bbox_normalize_stds'])[None, :] + np.array(cfg['bbox_normalize_means'])[None, :]
        boxes = np.hstack([bbox_helper.compute_loc_bboxes(rois[:, 1:5], delta), score[:, None]])
        for b in range(B):
            sel = np.where(rois[:, 0] == b)[0]
            s, bx = score[sel], boxes[sel]
            bx[:, :4] = bbox_helper.clip_bbox(bx[:, :4], image_info[b])
            if cfg['score_thresh'] > 0:
                ok = np.where(s > cfg['score_thresh'])[0]
                s, bx = s[ok], bx[ok]
            if s.size == 0:
                continue
            bx = bx[s.argsort()[::-1], :]
            keep = backend.nms(torch.from_numpy(bx).float(), cfg['nms_iou_thresh']).numpy()
            bx = bx[keep]
            found.append(np.hstack([np.full((bx.shape[0], 1), b), bx, np.full((bx.shape[0], 1), c)]))
    found = np.vstack(found)
    if cfg['top_n'] > 0:
        per_img = []
        for b in range(B):
            mine = found[found[:, 0] == b]
            per_img.append(mine[mine[:, -2].argsort()[::-1][:cfg['top_n']]])
        found = np.vstack(per_img)
    return torch.from_numpy(found).float().to(dev)
