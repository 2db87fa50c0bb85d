"""Validation pass of the detector -- the contract of validate() / validate_single()
(reference tools/faster_rcnn_train_val.py:773-884 / 886-981): run the model in eval mode over a loader, count the RPN
recall at IoU 0.5, write `results.txt.rank<r>` rows `image_id x1 y1 x2 y2 score class` (top 100 detections per
image, clipped, divided by the resize scale), then score them with utils.cal_mAP on rank 0.

The model forward is the HIP path (scda_amd.dropin.models...FasterRCNN_AdEx in eval mode: backbone + RPN + RoIPool + FC on
the MI355X, NMS through scda_nms_hip); this module is only the loop around it.
"""
import logging
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from scda_amd.dropin.utils import bbox_helper
from scda_amd.dropin.utils.cal_mAP import Cal_MAP

logger = logging.getLogger('global')


def _np(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def detection_rows(img_id, dts_per_image, gts_per_image, image_info_row, num_classes, resize_scale, coco=False):
    """rows of one image: 100 best by score, then class by class in that order (tools/faster_rcnn_train_val.py:838-858)"""
    rows = []
    order = dts_per_image[:, -2].argsort()[::-1][:100]
    dts_per_image = dts_per_image[order]
    for cls in range(1, num_classes):
        d = dts_per_image[dts_per_image[:, -1] == cls][:, 1:-1]
        d = bbox_helper.clip_bbox(d, image_info_row[:2])
        if len(d) > 0:
            d[:, :4] = d[:, :4] / resize_scale
        for bx in d:
            head = 'val2017/{0}.jpg'.format(img_id) if coco else '{0}'.format(img_id)
            rows.append('{0} {1} {2}\n'.format(head, ' '.join(map(str, bx)), cls))
    return rows


def validate(val_loader, model, cfg, results_dir, val_meta_file=None, dataset='cityscapes', device=None, score=True):
    """-> RPN recall (total recalled / total gts).  Loader items: (image [b,3,h,w], image_info [b,>=3], gts [b,G,5], ...,
    filenames).  Distributed when torch.distributed is initialised (each rank writes its own file, rank 0 scores after a
    one-element all-reduce, as the reference synchronises); single-process otherwise (validate_single)."""
    distributed = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(), dist.get_world_size()) if distributed else (0, 1)
    if device is None:
        device = next(model.parameters()).device
    was_training = model.training
    model.eval()
    num_classes = int(cfg['shared']['num_classes'])
    os.makedirs(results_dir, exist_ok=True)
    if rank == 0:   # stale files of a previous, wider run would be concatenated into results.txt
        for f in os.listdir(results_dir):
            if 'results.txt.rank' in f and int(f.split('k')[-1]) >= world:
                logger.info("remove %s" % f)
                os.remove(os.path.join(results_dir, f))
    total_rc = total_gt = 0
    with open(os.path.join(results_dir, 'results.txt.rank%d' % rank), 'w') as fout, torch.no_grad():
        for it, item in enumerate(val_loader):
            img, img_info, gt_boxes, filenames = item[0], item[1], item[2], item[-1]
            t0 = time.time()
            x = {'cfg': cfg, 'image': img.to(device, non_blocking=True), 'image_info': img_info,
                 'ground_truth_bboxes': gt_boxes, 'ignore_regions': None}
            outputs = model(x)['predict']
            proposals, bboxes = _np(outputs[0]), _np(outputs[1])
            t1 = time.time()
            gts_np, info_np = _np(gt_boxes), _np(img_info)
            for b in range(img.shape[0]):
                img_id = filenames[b].rsplit('/', 1)[-1].rsplit('.', 1)[0]
                scale = info_np[b, 2] if dataset == 'coco' else info_np[b, -1]
                rc, ng = bbox_helper.compute_recall(proposals[proposals[:, 0] == b][:, 1:5], gts_np[b])
                total_rc += rc
                total_gt += ng
                fout.writelines(detection_rows(img_id, bboxes[bboxes[:, 0] == b], gts_np[b], info_np[b], num_classes, scale,
                                               coco=(dataset == 'coco')))
                fout.flush()
            logger.info('Test: [%d/%d] Time: %.3f %d/%d' % (it, len(val_loader), t1 - t0, total_rc, total_gt))
    logger.info('rpn300 recall=%f' % (total_rc / total_gt))
    if distributed:
        sync = torch.ones(1, device=device)
        dist.all_reduce(sync)
    if score and rank == 0:
        if dataset == 'coco':
            raise NotImplementedError("COCO scoring needs pycocotools (datasets/pycocotools in the reference); "
                                      "results.txt.rank* are written, score them there")
        Cal_MAP(results_dir, val_meta_file, num_classes)
    if was_training:
        model.train()
    return total_rc / total_gt
