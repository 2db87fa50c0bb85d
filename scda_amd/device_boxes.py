"""Device-resident box logic of the detector: anchor labelling and RPN proposal generation on the
MI355X (scda_amd/csrc/box_ops.hip), replacing the numpy passes of functions/anchor_target.py:38-107 and
functions/rpn_proposal.py:36-66 for batch-1 device inputs.

What stays on the host, because it IS the reference's observable behaviour:
  * random draws from numpy's global generator (np.random.choice for the surplus positives / negatives): the device sends
    back two counts, the host draws exactly what the reference would and uploads the indices to drop;
  * the ranking of RPN scores (np.argpartition + np.argsort -- the order of tied scores is numpy's) and np.exp of the ranked
    candidates' size deltas (numpy's float32 exp is its own, not correctly rounded, SIMD routine): the host receives the RPN
    outputs, ranks, exponentiates 2 x 12000 numbers and uploads order + factors.
What no longer crosses PCIe or touches numpy: the 30720 x G IoU matrix, the 30720 x 9 target maps, the decoded / clipped / filtered
candidate boxes, the NMS input and the gather of the kept rows.
All kernels here run on the box logic's high-priority side stream (their inputs do not depend on the compute stream's
backlog), so the host's waits stay short.

`enabled()` is the switch (SCDA_DEVICE_BOXES=0 restores the numpy passes, e.g. for an A/B of host CPU time)."""
import os

import numpy as np
import torch

from scda_amd import native as N
from scda_amd.dropin import backend
from scda_amd.dropin.utils import anchor_helper

_ANCHORS = {}
_BUFS = {}


def enabled():
    return os.environ.get("SCDA_DEVICE_BOXES", "1") != "0" and torch.cuda.is_available()


def anchors_on_device(fh, fw, cfg, dev):
    key = (fh, fw, tuple(cfg['anchor_scales']), cfg['anchor_stride'], dev.index)
    hit = _ANCHORS.get(key)
    if hit is None:
        a = anchor_helper.get_anchors_over_plane(fh, fw, cfg['anchor_ratios'], cfg['anchor_scales'], cfg['anchor_stride'])
        a64 = torch.from_numpy(np.array(a, dtype=np.float64)).to(dev)
        a32 = torch.from_numpy(a.astype(np.float32)).to(dev)      # the cast the reference applies before the IoU (bbox_helper.py:9)
        hit = _ANCHORS[key] = (a32, a64)
    return hit


def _bufs(KA, G, dev):
    key = (KA, dev.index)
    b = _BUFS.get(key)
    if b is None or b["gt_best"].numel() < G:
        i32 = dict(dtype=torch.int32, device=dev)
        b = _BUFS[key] = {"best_iou": torch.empty(KA, dtype=torch.float32, device=dev), "best_gt": torch.empty(KA, **i32),
                          "gt_best": torch.empty(max(G, 64), **i32), "labels": torch.empty(KA, dtype=torch.int8, device=dev),
                          "pos_list": torch.empty(KA, **i32), "neg_list": torch.empty(KA, **i32), "counts": torch.zeros(2, **i32),
                          "counts_host": torch.zeros(2, dtype=torch.int32, pin_memory=True)}
    return b


def _pinned_i32(a):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32))
    p = torch.empty(t.shape, dtype=torch.int32, pin_memory=True)
    p.copy_(t)
    return p


def anchor_targets(feature_size, cfg, gts_dev):
    """functions/anchor_target.py:16-116 for ONE image whose ground truth lives on the device.
    -> cls_targets int64 [1,A,fh,fw], loc_targets, loc_masks fp32 [1,4A,fh,fw], normaliser (int)"""
    B, A4, fh, fw = feature_size
    A = A4 // 4
    assert B == 1 and A * 4 == A4
    dev = gts_dev.device
    a32, a64 = anchors_on_device(fh, fw, cfg, dev)
    KA = a32.shape[0]
    main = torch.cuda.current_stream(dev)
    aux = backend._aux_stream(dev)
    host_gts = getattr(gts_dev, "_scda_host", None)
    with torch.cuda.stream(aux):
        if host_gts is not None:      # re-upload the (tiny) boxes on the side stream: no dependence on the compute stream at all
            pin = torch.empty(host_gts.shape[1:], dtype=torch.float32, pin_memory=True)
            pin.copy_(torch.from_numpy(np.ascontiguousarray(host_gts[0], dtype=np.float32)))
            gts = pin.to(dev, non_blocking=True)
        else:
            aux.wait_stream(main)
            gts = gts_dev[0].contiguous()
        G = gts.shape[0]
        bufs = _bufs(KA, G, dev)
        N.anchor_label(a32, gts, cfg['negative_iou_thresh'], cfg['positive_iou_thresh'], 0.1, bufs)
        bufs["counts_host"].copy_(bufs["counts"], non_blocking=True)
        aux.synchronize()
        n_pos, n_neg = (int(v) for v in bufs["counts_host"])
        # sub-sampling: the reference's two np.random.choice calls, on the counts the device found (:66-80)
        budget = cfg['rpn_batch_size'] * B
        max_pos = int(cfg['positive_percent'] * budget)
        drop_pos = drop_neg = None
        if n_pos > max_pos:
            drop_pos = _pinned_i32(np.random.choice(n_pos, size=n_pos - max_pos, replace=False)).to(dev, non_blocking=True)
            n_pos = max_pos
        max_neg = budget - n_pos
        if n_neg > max_neg:
            drop_neg = _pinned_i32(np.random.choice(n_neg, size=n_neg - max_neg, replace=False)).to(dev, non_blocking=True)
            n_neg = max_neg
        cls_t, loc_t, loc_m = N.anchor_finalize(bufs, drop_pos, drop_neg, a64, gts, A, fh, fw)
    main.wait_stream(aux)
    for t in (cls_t, loc_t, loc_m):
        t.record_stream(main)
    return cls_t, loc_t, loc_m, max(1, n_pos + n_neg)


def rpn_proposals(prob_dev, loc_dev, cfg, image_info, scores_host=None, loc_host=None):
    """functions/rpn_proposal.py:17-74 with the RPN outputs resident on the device.  prob_dev [B,2A,fh,fw] (soft-maxed),
    loc_dev [B,4A,fh,fw]; scores_host / loc_host: CPU copies if the caller already has them (else they are fetched here).
    -> CPU float tensor [N,6] (b,x1,y1,x2,y2,score), as the reference returns; `._scda_dev` = the same rows on the device"""
    B, A4, fh, fw = loc_dev.shape
    A = A4 // 4
    dev = loc_dev.device
    _, a64 = anchors_on_device(fh, fw, cfg, dev)
    KA = fh * fw * A
    if scores_host is None:
        scores_host = prob_dev.detach().cpu()
    if loc_host is None:
        loc_host = loc_dev.detach().cpu()
    cls = scores_host.permute(0, 2, 3, 1).contiguous().view(B, KA, -1).numpy()
    loc = loc_host.permute(0, 2, 3, 1).contiguous().view(B, KA, 4).numpy()
    info = image_info.cpu().numpy() if torch.is_tensor(image_info) else np.asarray(image_info)
    top_n, keep_n = cfg['pre_nms_top_n'], cfg['post_nms_top_n']
    main = torch.cuda.current_stream(dev)
    aux = backend._aux_stream(dev)
    outs = []
    # No stream dependency is needed: the host holds a copy of the RPN outputs (its own .cpu() above, or the caller's event-
    # synchronised pinned copy), so the kernels that produced prob_dev / loc_dev have completed.  (Waiting for the compute
    # stream here would put the proposals behind everything queued since -- e.g. the whole target-image backbone.)
    with torch.cuda.stream(aux):
        for b in range(B):
            score = cls[b, :, -1]
            if top_n <= 0 or top_n > score.shape[0]:
                order = score.argsort()[::-1]
            else:
                cand = np.argpartition(-score, top_n)[:top_n]
                order = cand[np.argsort(-score[cand])]
            order_dev = _pinned_i32(order).to(dev, non_blocking=True)
            ewh = torch.from_numpy(np.exp(loc[b, order, 2:4]))           # float32 in, float32 out: numpy's own exp (bbox_helper.py:97)
            ewh_pin = torch.empty(ewh.shape, dtype=torch.float32, pin_memory=True)
            ewh_pin.copy_(ewh)
            out6, num = N.proposals_from_ranking(order_dev, ewh_pin.to(dev, non_blocking=True), a64, loc_dev[b].contiguous(), prob_dev[b].contiguous(), A, fh, fw,
                                                 float(info[b][0]), float(info[b][1]), float(cfg['roi_min_size']),
                                                 float(cfg['nms_iou_thresh']), max(keep_n, 0), float(b))
            host = torch.empty(out6.numel() + 1, dtype=torch.float32, pin_memory=True)
            host[:-1].copy_(out6.view(-1), non_blocking=True)
            host[-1:].copy_(num.to(torch.float32), non_blocking=True)
            outs.append((host, out6))
        aux.synchronize()
    rows, devs = [], []
    for host, out6 in outs:
        k = int(host[-1])
        rows.append(host[:-1].view(-1, 6)[:k].clone())
        devs.append(out6[:k])
    res = torch.cat(rows, 0) if len(rows) > 1 else rows[0]
    res._scda_dev = torch.cat(devs, 0) if len(devs) > 1 else devs[0]
    return res


_PT_BUFS = {}


def _pt_bufs(n, dev):
    b = _PT_BUFS.get(dev.index)
    if b is None or b["cap"] < n:
        cap = max(4096, n)
        i32 = dict(dtype=torch.int32, device=dev)
        b = _PT_BUFS[dev.index] = {"cap": cap, "rois": torch.empty(cap, 4, dtype=torch.float32, device=dev),
                                   "best_iou": torch.empty(cap, dtype=torch.float32, device=dev), "best_gt": torch.empty(cap, **i32),
                                   "labels": torch.empty(cap, dtype=torch.int8, device=dev), "pos_list": torch.empty(cap, **i32),
                                   "neg_list": torch.empty(cap, **i32), "counts": torch.zeros(2, **i32),
                                   "host": torch.zeros(3 * cap + 2, dtype=torch.int32, pin_memory=True)}
    return b


def proposal_targets_legal(cfg, gts, ignore_regions, use_ohem):
    return (ignore_regions is None and not use_ohem and cfg['append_gts'] and cfg['positive_iou_thresh'] >= cfg['negative_iou_thresh_hi']
            and torch.is_tensor(gts) and gts.is_cuda and gts.dtype == torch.float32 and gts.shape[0] == 1 and enabled())


def proposal_targets(proposals, cfg, gts_dev, image_info):
    """functions/proposal_target.py:17-177 for ONE image with the candidates resident on the device: clip, the (N + G) x G IoU, best gt,
    the threshold tests and the ordered foreground / background index lists are kernels (box_ops.hip: scda_proposal_match_hip), the
    gather of the sampled rows and the [512, 4C] target / weight maps too (scda_proposal_finalize_hip).  The host keeps what IS the
    reference's observable behaviour: the order in which Python-set arithmetic leaves the negatives (:87,90), the np.random.choice
    draws (:95-109, :149-151) and numpy's float32 log on the <= 128 foreground rows (utils/bbox_helper.py:70-86).
    -> rois fp32 [R,5], labels int64 [R], loc_targets, loc_weights fp32 [R,4C] on the device; None when the image has no usable
    ground truth (the caller's numpy path then behaves as the reference does)."""
    from scda_amd.dropin.functions import proposal_target as PT
    from scda_amd.dropin.utils import bbox_helper
    dev = gts_dev.device
    host_gts = getattr(gts_dev, "_scda_host", None)
    gts = (host_gts if host_gts is not None else gts_dev.detach().cpu().numpy())[0]
    gts = gts[(gts[:, 2] > gts[:, 0] + 1) & (gts[:, 3] > gts[:, 1] + 1)]           # drop zero-padded gt rows (:40-41)
    props_host = proposals.numpy() if torch.is_tensor(proposals) else np.asarray(proposals)
    props_host = props_host[props_host[:, 0] == 0]
    info = backend.host_array(image_info)
    if gts.shape[0] == 0 or props_host.shape[0] + gts.shape[0] == 0:
        return None
    gts = np.ascontiguousarray(gts, dtype=np.float32)
    n_prop, G = props_host.shape[0], gts.shape[0]
    n = n_prop + G
    C, per_image = cfg['num_classes'], cfg['batch_size']
    main = torch.cuda.current_stream(dev)
    aux = backend._aux_stream(dev)
    props_dev = getattr(proposals, "_scda_dev", None)
    if props_dev is not None and (props_dev.shape[0] != n_prop or props_dev.device != dev):
        props_dev = None
    with torch.cuda.stream(aux):
        pin = torch.empty(gts.shape, dtype=torch.float32, pin_memory=True)
        pin.copy_(torch.from_numpy(gts))
        g_dev = pin.to(dev, non_blocking=True)
        if props_dev is None:
            pp = torch.empty(props_host.shape, dtype=torch.float32, pin_memory=True)
            pp.copy_(torch.from_numpy(np.ascontiguousarray(props_host, dtype=np.float32)))
            props_dev = pp.to(dev, non_blocking=True)
        bufs = _pt_bufs(n, dev)
        N.proposal_match(props_dev.contiguous(), g_dev, float(info[0][0]), float(info[0][1]), cfg['positive_iou_thresh'],
                         cfg['negative_iou_thresh_hi'], cfg['negative_iou_thresh_lo'], bufs)
        cap, host = bufs["cap"], bufs["host"]
        host[0:n].copy_(bufs["pos_list"][:n], non_blocking=True)
        host[cap:cap + n].copy_(bufs["neg_list"][:n], non_blocking=True)
        host[2 * cap:2 * cap + n].copy_(bufs["best_gt"][:n], non_blocking=True)
        host[3 * cap:].copy_(bufs["counts"], non_blocking=True)
        aux.synchronize()
        hv = host.numpy()
        n_pos, n_neg = int(hv[3 * cap]), int(hv[3 * cap + 1])
        # np.where(...)[0] of the reference: ascending int64 index arrays (:55-59); np.unique leaves the positives as they are
        pos_r = hv[0:n_pos].astype(np.int64)
        pos_g = hv[2 * cap:2 * cap + n].astype(np.int64)[pos_r]
        neg_r = hv[cap:cap + n_neg].astype(np.int64)
        neg_r = np.array(list(set(neg_r) - set(pos_r)))                               # :90 -- CPython's set order, kept verbatim
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
        # the host's own copy of the clipped candidates: the <= 128 foreground rows for the encode, the sampled rows for `_scda_host`
        cand = bbox_helper.clip_bbox(np.vstack([props_host[:, 1:5], gts[:, :4]]), info[0])
        pos_rois, pos_gts = cand[pos_r], gts[pos_g]
        n_pos, n_neg = len(pos_r), len(neg_r)
        m = n_pos + n_neg
        enc = bbox_helper.compute_loc_targets(pos_rois, pos_gts)
        if cfg['bbox_normalize_stats_precomputed']:
            enc = (enc - np.array(cfg['bbox_normalize_means'])[None, :]) / np.array(cfg['bbox_normalize_stds'])[None, :]
        sel = np.array(pos_r + neg_r, dtype=np.int64)
        gt_of = np.concatenate([np.array(pos_g, dtype=np.int64), np.full(n_neg, -1, dtype=np.int64)])
        enc_all = np.zeros((m, 4), dtype=np.float32)
        enc_all[:n_pos] = enc.astype(np.float32)          # the reference's float64 targets, cast where it casts (`.float()`, :176)
        if m < per_image:                                   # pad by resampling (with replacement) what we have (:149-155)
            again = np.random.choice(m, size=per_image - m, replace=True)
            sel, gt_of, enc_all = np.concatenate([sel, sel[again]]), np.concatenate([gt_of, gt_of[again]]), np.vstack([enc_all, enc_all[again]])
        R = sel.shape[0]
        up = torch.empty(R, 6, dtype=torch.int32, pin_memory=True)
        up[:, 0].copy_(torch.from_numpy(sel.astype(np.int32)))
        up[:, 1].copy_(torch.from_numpy(gt_of.astype(np.int32)))
        up[:, 2:].copy_(torch.from_numpy(enc_all).view(torch.int32))
        up_dev = up.to(dev, non_blocking=True)
        rois, labels, t, w = N.proposal_finalize(bufs["rois"], up_dev[:, 0].contiguous(), up_dev[:, 1].contiguous(),
                                                 up_dev[:, 2:].contiguous().view(torch.float32), g_dev, C, 0.0)
    main.wait_stream(aux)
    for x in (rois, labels, t, w):
        x.record_stream(main)
    rois_np = np.hstack([np.zeros((R, 1), dtype=np.float32), cand[sel]]).astype(np.float32)
    rois._scda_host = rois_np          # lets the cluster-region generator read the RoIs without a device->host copy
    n_fg = int((gt_of >= 0).sum())     # labels > 0 <=> foreground: class ids start at 1
    PT.history[0] += n_fg
    PT.history[1] += R - n_fg
    return rois, labels, t, w
