"""utils.cal_mAP -- Cityscapes-style detection mAP, the contract of the reference's utils/cal_mAP.py.

Host-side bookkeeping over text lists (a few thousand rows per validation pass), so it stays on the CPU; the matching
is done per class with the IoUs of one detection against all ground truths of its image computed as one numpy
expression.  Everything a caller of the reference can observe is kept:
  * results rows `name x1 y1 x2 y2 score label`, coordinates truncated with int(float(.))        (cal_mAP.py:49-66)
  * IoU with the +1 pixel convention, only for boxes that STRICTLY overlap in both axes; the first ground truth with
    the largest IoU wins; no overlap at all -> (-1, -1)                                         (cal_mAP.py:68-93)
  * detections visited by descending score, ties in list order; a ground truth is claimed once (cal_mAP.py:95-116)
  * AP = sum_v (rec[v] - rec[v-1]) * max_{u>=v} prec[u]; a class without ground truth divides by zero (nan/inf, with
    numpy's warning), a class without detections raises ValueError as np.max of an empty array does (cal_mAP.py:117-132)
"""
import glob
import logging
import os
from collections import defaultdict

import numpy as np

logger = logging.getLogger('global')


def parse_gts(gts_list, num_classes):
    """val meta list -> {'num': per-class count, name: {'height','width','bbox_num','bbox': {cls: [[x1,y1,x2,y2]..]},
    'is_det': {cls: zeros}}}.  Records start at a '#' line: +1 path, +3 height, +4 width, +7 box count, +8.. boxes
    as `label x1 y1 x2 y2`."""
    gts = defaultdict(list)
    gts['num'] = np.zeros(num_classes)
    for at, line in enumerate(gts_list):
        if not line.startswith('#'):
            continue
        name = gts_list[at + 1].strip().split('/')[-1][0:-4]
        count = int(gts_list[at + 7])
        rec = defaultdict(list)
        rec['height'] = gts_list[at + 3].strip()
        rec['width'] = gts_list[at + 4].strip()
        rec['bbox_num'] = count
        rec['bbox'] = defaultdict(list)
        for row in gts_list[at + 8: at + 8 + count]:
            f = row.split()
            label = int(f[0])
            rec['bbox'][label].append([int(f[1]), int(f[2]), int(f[3]), int(f[4])])
            gts['num'][label] += 1
        rec['is_det'] = defaultdict(list)
        for c in range(1, num_classes):
            rec['is_det'][c] = np.zeros(len(rec['bbox'][c]))
        gts[name] = rec
    return gts


def parse_res(res_list):
    """results rows -> {label: [[x1, y1, x2, y2, score, name], ...]} in file order"""
    out = defaultdict(list)
    for row in res_list:
        f = row.split()
        out[int(f[6])].append([int(float(f[1])), int(float(f[2])), int(float(f[3])), int(float(f[4])), float(f[5]), f[0]])
    return out


def calIoU(result, gt_i):
    """best (IoU, index) of one detection over the ground truths `gt_i` of its image and class"""
    if len(gt_i) == 0:
        return -1, -1
    g = np.asarray(gt_i, dtype=np.int64).reshape(-1, 4)
    x1, y1, x2, y2 = (int(v) for v in result[:4])
    ix1, iy1 = np.maximum(g[:, 0], x1), np.maximum(g[:, 1], y1)
    ix2, iy2 = np.minimum(g[:, 2], x2), np.minimum(g[:, 3], y2)
    hit = (ix1 < ix2) & (iy1 < iy2)
    if not hit.any():
        return -1, -1
    inter = (ix2 - ix1 + 1) * (iy2 - iy1 + 1)
    union = (x2 - x1 + 1) * (y2 - y1 + 1) + (g[:, 2] - g[:, 0] + 1) * (g[:, 3] - g[:, 1] + 1) - inter
    iou = np.where(hit, inter / union, -np.inf)
    k = int(np.argmax(iou))                    # first maximum, as a strict `>` scan keeps
    return float(iou[k]), k


def cal_mAP(gts, results, num_classes, overlap_thre):
    ap = np.zeros(num_classes)
    max_recall = np.zeros(num_classes)
    for c in range(1, num_classes):
        dets = sorted(results[c], key=lambda d: d[4], reverse=True)     # stable: ties keep list order
        n = len(dets)
        total = gts['num'][c]
        logger.info('sum_gt: {}'.format(total))
        hit = np.zeros(n)
        for k, d in enumerate(dets):
            img = gts[d[-1]]
            best, which = calIoU(d, img['bbox'][int(c)])
            if best >= overlap_thre and img['is_det'][c][which] == 0:
                hit[k] = 1
                img['is_det'][c][which] = 1
        tp = np.cumsum(hit)
        fp = np.cumsum(1.0 - hit)
        rec = tp / total
        prec = tp / (tp + fp)
        env = np.maximum.accumulate(prec[::-1])[::-1] if n else prec
        steps = np.diff(rec, prepend=0.0) if n else rec
        a = 0.0
        for v in range(n):                     # left-to-right sum, the order the reference accumulates in
            a += (rec[0] if v == 0 else steps[v]) * env[v]
        ap[c] = a
        max_recall[c] = np.max(rec)            # ValueError for a class with no detections, as in the reference
        logger.info('class {} --- ap: {}   max recall: {}'.format(c, ap[c], max_recall[c]))
    return ap, max_recall


def Cal_MAP1(res_list, gts_list, num_classes):
    """lists of lines in, mAP over classes 1.. out (cal_mAP.py:136-152)"""
    num_classes = int(num_classes)
    ap, max_recall = cal_mAP(parse_gts(gts_list, num_classes), parse_res(res_list), num_classes, 0.5)
    mAP = np.mean(ap[1:])
    logger.info('mAP: {}   max recall: {}'.format(mAP, np.mean(max_recall[1:])))
    return mAP


def Cal_MAP(res_dir, gts_list, num_classes):
    """concatenate <res_dir>/results.txt.rank* into results.txt, score it against the meta file `gts_list`, print
    (cal_mAP.py:154-174; the reference shells out to `cat`, the file it leaves behind is the same)"""
    parts = sorted(glob.glob(os.path.join(res_dir, 'results.txt.rank*')))
    rows = []
    for p in parts:
        with open(p, 'r', encoding='utf-8') as f:
            rows += f.readlines()
    with open(os.path.join(res_dir, 'results.txt'), 'w', encoding='utf-8') as f:
        f.writelines(rows)
    with open(gts_list, 'r', encoding='utf-8') as f:
        meta = f.readlines()
    num_classes = int(num_classes)
    ap, max_recall = cal_mAP(parse_gts(meta, num_classes), parse_res(rows), num_classes, 0.5)
    mAP, m_rec = np.mean(ap[1:]), np.mean(max_recall[1:])
    print('--------------------')
    print('mAP: {}   max recall: {}'.format(mAP, m_rec))
    print('--------------------')
    return mAP
