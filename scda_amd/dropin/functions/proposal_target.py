"""RCNN training targets -- the contract of functions/proposal_target.py:17-177: append gts, match by IoU, sample
fg/bg to exactly `batch_size` RoIs per image, class-specific normalised regression targets.

Two details are part of the observable result and are kept verbatim: the negatives are de-duplicated through Python
`set` arithmetic (their order, and therefore what np.random.choice picks, follows CPython's set iteration:
functions/proposal_target.py:87,90), and images that come up short are padded by resampling with replacement
(:149-155)."""
import logging

import numpy as np
import torch

from scda_amd.dropin import backend
from scda_amd.dropin.utils import bbox_helper

logger = logging.getLogger('global')
history = [0, 0]


def _np(x):
    return backend.host_array(x)


def compute_proposal_targets(proposals, cfg, ground_truth_bboxes, image_info, ignore_regions=None, use_ohem=False):
    """proposals [N,>=5] (b,x1,y1,x2,y2,..) -> rois fp32 [R,5], labels int64 [R], loc_targets, loc_weights fp32 [R,4C]"""
    dev = ground_truth_bboxes.device if torch.is_tensor(ground_truth_bboxes) else torch.device('cpu')
    if dev.type == 'cuda':
        from scda_amd import device_boxes
        if device_boxes.proposal_targets_legal(cfg, ground_truth_bboxes, ignore_regions, use_ohem):
            out = device_boxes.proposal_targets(proposals, cfg, ground_truth_bboxes, image_info)   # IoU / matching / gather on the MI355X
            if out is not None:
                return out
    proposals, gts_all, image_info, ignore_regions = map(_np, (proposals, ground_truth_bboxes, image_info, ignore_regions))
    C = cfg['num_classes']
    per_image = cfg['batch_size']
    acc_rois, acc_labels, acc_t, acc_w = [], [], [], []
    for b in range(gts_all.shape[0]):
        rois = proposals[proposals[:, 0] == b][:, 1:5]
        gts = gts_all[b]
        gts = gts[(gts[:, 2] > gts[:, 0] + 1) & (gts[:, 3] > gts[:, 1] + 1)]  # drop zero-padded gt rows
        if cfg['append_gts']:
            rois = np.vstack([rois, gts[:, :4]])
        rois = bbox_helper.clip_bbox(rois, image_info[b])
        if rois.shape[0] == 0 or gts.shape[0] == 0:
            continue
        iou = bbox_helper.bbox_iou_overlaps(rois, gts)
        best_gt, best_iou = iou.argmax(axis=1), iou.max(axis=1)

        pos_r = np.where(best_iou > cfg['positive_iou_thresh'])[0]
        pos_g = best_gt[pos_r]
        pos_r, first = np.unique(pos_r, return_index=True)
        pos_g = pos_g[first]
        neg_r = np.where((best_iou < cfg['negative_iou_thresh_hi']) & (best_iou >= cfg['negative_iou_thresh_lo']))[0]
        if ignore_regions is not None:
            ign = ignore_regions[b]
            ign = ign[ign[:, 2] - ign[:, 0] > 1]
            if ign.shape[0] > 0:
                iof = bbox_helper.bbox_iof_overlaps(rois, ign)
                inside = np.where(iof.max(axis=1) > cfg['ignore_iou_thresh'])[0]
                neg_r = np.array(list(set(neg_r) - set(inside)))
        neg_r = np.array(list(set(neg_r) - set(pos_r)))

        n_pos = len(pos_r)
        if not use_ohem:
            want_pos = int(cfg['positive_percent'] * per_image)
            if want_pos < n_pos:
                pick = np.random.choice(n_pos, size=want_pos, replace=False)
                pos_r, pos_g = pos_r[pick], pos_g[pick]
                n_pos = want_pos
            want_neg = per_image - n_pos
            if want_neg < len(neg_r):
                pick = np.random.choice(len(neg_r), size=want_neg, replace=False)
                neg_r = neg_r[pick]

        pos_r, pos_g, neg_r = list(pos_r), list(pos_g), list(neg_r)
        pos_rois, pos_gts, neg_rois = rois[pos_r], gts[pos_g], rois[neg_r]
        sampled = np.vstack([pos_rois, neg_rois])
        n_pos, n_neg = pos_rois.shape[0], neg_rois.shape[0]
        n = n_pos + n_neg

        pos_labels = pos_gts[:, 4].astype(np.int32)
        labels = np.concatenate([pos_labels, np.zeros(n_neg)]).astype(np.int32)
        t = np.zeros([n, C, 4])
        w = np.zeros([n, C, 4])
        enc = bbox_helper.compute_loc_targets(pos_rois, pos_gts)
        if cfg['bbox_normalize_stats_precomputed']:
            enc = (enc - np.array(cfg['bbox_normalize_means'])[None, :]) / np.array(cfg['bbox_normalize_stds'])[None, :]
        t[range(n_pos), pos_labels, :] = enc
        w[range(n_pos), pos_labels, :] = 1
        t, w = t.reshape([n, -1]), w.reshape([n, -1])
        sampled = np.hstack([np.full((n, 1), b), sampled])

        if n < per_image:  # pad by resampling (with replacement) what we have
            again = np.random.choice(n, size=per_image - n, replace=True)
            sampled = np.vstack([sampled, sampled[again]])
            labels = np.concatenate([labels, labels[again]])
            t = np.vstack([t, t[again]])
            w = np.vstack([w, w[again]])
        acc_rois.append(sampled); acc_labels.append(labels); acc_t.append(t); acc_w.append(w)

    all_labels = np.concatenate(acc_labels)
    n_fg = int((all_labels > 0).sum())
    history[0] += n_fg
    history[1] += all_labels.shape[0] - n_fg
    logger.debug('proposal_target(pos/neg): %d=%d+%d' % (all_labels.shape[0], n_fg, all_labels.shape[0] - n_fg))

    def dev_t(a, kind):
        x = torch.from_numpy(a)
        x = x.float() if kind == 'f' else x.long()
        if dev.type == 'cuda':
            from scda_amd import native
            return native.upload(x, dev)
        return x.to(dev).contiguous()

    rois_np = np.vstack(acc_rois).astype(np.float32)
    rois_dev = dev_t(rois_np, 'f')
    rois_dev._scda_host = rois_np   # lets the cluster-region generator read the RoIs without a device->host copy
    return rois_dev, dev_t(all_labels, 'l'), dev_t(np.vstack(acc_t), 'f'), dev_t(np.vstack(acc_w), 'f')
