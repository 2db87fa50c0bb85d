"""overlap(bboxes1 np[N,>=4], bboxes2 np[M,>=4]) -> np[N,M]: extensions/_bbox_helper/bbox_helper.py:5-15
(IoU without the +1 convention, union clamped to >= 1)."""
import numpy as np
import torch

from scda_amd import native as N


def overlap(bboxes1, bboxes2):
    dev = torch.device("cuda", torch.cuda.current_device())
    b1 = torch.from_numpy(np.ascontiguousarray(bboxes1[:, :4])).float().to(dev).contiguous()
    b2 = torch.from_numpy(np.ascontiguousarray(bboxes2[:, :4])).float().to(dev).contiguous()
    return N.iou_overlaps(b1, b2).cpu().numpy()
