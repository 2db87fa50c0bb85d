"""nms(dets, thresh) -> CPU LongTensor, the contract of extensions/_nms/pth_nms.py:5-47:
`dets` [N,5] float (x1,y1,x2,y2,score) ALREADY sorted by score; returns indices into it,
on the CPU (callers do `.numpy()`: functions/rpn_proposal.py:64, functions/predict_bbox.py:49).
Mask kernel and greedy sweep both run on the MI355X; only the keep list crosses PCIe."""
from scda_amd.dropin import backend


def pth_nms(dets, thresh):
    if dets.dim() != 2 or dets.shape[1] != 5:
        raise ValueError("nms expects dets of shape [N,5]")
    return backend.nms(dets, thresh)
