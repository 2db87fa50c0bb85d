"""RPN proposal generation -- the contract of functions/rpn_proposal.py:17-74: top-k by objectness, decode,
clip, min-size filter, NMS, keep post_nms_top_n.  Ordering uses the same numpy selection calls as the reference
(np.argpartition / np.argsort tie behaviour is part of the result); NMS + greedy sweep run on the MI355X and stop
as soon as post_nms_top_n boxes are kept."""
import numpy as np
import torch

from scda_amd.dropin import backend
from scda_amd.dropin.utils import anchor_helper, bbox_helper


# Parity tests (scda_amd/probe.py `rpn_output`: callable(conv_cls, conv_loc) -> (conv_cls, conv_loc)).  The ranking below is a
# discontinuous function of the scores: among 30720 fp32 soft-max outputs some pairs sit closer than the 1e-6 by which two
# correct implementations differ, and ONE swapped pair changes which RoIs get sampled downstream.  The full-size parity test
# records the CPU oracle's RPN outputs and hands them to the device run here (after asserting agreement to 1e-5), so that
# everything after this point is compared on identical discrete decisions (tests/model_common.py).  None outside a probed step.
from scda_amd.probe import rpn_output as _probe_rpn_output


def compute_rpn_proposals(conv_cls, conv_loc, cfg, image_info):
    """conv_cls [B, A*2, h, w] (soft-maxed), conv_loc [B, A*4, h, w] -> CPU float tensor [N,6] (b,x1,y1,x2,y2,score)"""
    host = getattr(conv_cls, "_scda_host", None)    # () -> CPU copies (cls, loc) the caller already started, or None
    on_device = conv_loc.is_cuda
    cls_host = loc_host = None
    if host is not None:
        cls_host, loc_host = host()
    hook = _probe_rpn_output()
    if hook is not None:
        dev = conv_loc.device
        cls_host, loc_host = hook(cls_host if cls_host is not None else conv_cls,
                                             loc_host if loc_host is not None else conv_loc)
        if cls_host.is_cuda:
            cls_host = loc_host = None
        elif on_device:                              # the hook handed back host tensors: the device path continues on their upload
            conv_cls, conv_loc = cls_host.to(dev), loc_host.to(dev)
        else:
            conv_cls, conv_loc = cls_host, loc_host
    if on_device:
        from scda_amd import device_boxes
        if device_boxes.enabled():     # decode / clip / size test / NMS / gather on the MI355X; the host ranks and exponentiates
            return device_boxes.rpn_proposals(conv_cls, conv_loc, cfg, image_info, cls_host, loc_host)
    B, A4, fh, fw = conv_loc.shape
    A = A4 // 4
    assert A * 4 == A4
    anchors = anchor_helper.get_anchors_over_plane(fh, fw, cfg['anchor_ratios'], cfg['anchor_scales'], cfg['anchor_stride'])
    KA = fh * fw * A
    cls = conv_cls.permute(0, 2, 3, 1).contiguous().view(B, KA, -1).cpu().numpy()
    loc = conv_loc.permute(0, 2, 3, 1).contiguous().view(B, KA, 4).cpu().numpy()
    if torch.is_tensor(image_info):
        image_info = image_info.cpu().numpy()

    top_n, keep_n = cfg['pre_nms_top_n'], cfg['post_nms_top_n']
    out = []
    for b in range(B):
        score = cls[b, :, -1]
        if top_n <= 0 or top_n > score.shape[0]:
            order = score.argsort()[::-1]
        else:
            cand = np.argpartition(-score, top_n)[:top_n]
            order = cand[np.argsort(-score[cand])]
        score = score[order]
        boxes = bbox_helper.compute_loc_bboxes(anchors[order, :], loc[b, order, :])
        boxes = bbox_helper.clip_bbox(boxes, image_info[b])
        props = np.hstack([boxes, score[:, None]])
        big = (props[:, 2] - props[:, 0] + 1 >= cfg['roi_min_size']) & (props[:, 3] - props[:, 1] + 1 >= cfg['roi_min_size'])
        props = props[big]
        keep = backend.nms(torch.from_numpy(props).float(), cfg['nms_iou_thresh'], max_keep=max(keep_n, 0)).numpy()
        if keep_n > 0:
            keep = keep[:keep_n]
        props = props[keep]
        out.append(np.hstack([np.full((len(keep), 1), b), props]))
    res = torch.from_numpy(np.vstack(out)).float()
    if res.dim() < 2:
        res = res.unsqueeze(0)
    return res
