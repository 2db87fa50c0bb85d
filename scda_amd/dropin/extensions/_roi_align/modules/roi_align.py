"""RoIAlign / RoIAlignAvg / RoIAlignMax modules -- extensions/_roi_align/modules/roi_align.py:6-44.
Avg/Max sample an (A+1)^2 grid and reduce 2x2 windows with stride 1."""
import torch
from torch.nn.modules.module import Module

from scda_amd.dropin.extensions._roi_align.functions.roi_align import RoIAlignFunction


class RoIAlign(Module):
    def __init__(self, aligned_height, aligned_width, spatial_scale):
        super().__init__()
        self.aligned_width = int(aligned_width)
        self.aligned_height = int(aligned_height)
        self.spatial_scale = float(spatial_scale)

    def forward(self, features, rois):
        return RoIAlignFunction(self.aligned_height, self.aligned_width, self.spatial_scale)(features, rois)


def _win2x2(x):
    return x[:, :, :-1, :-1], x[:, :, :-1, 1:], x[:, :, 1:, :-1], x[:, :, 1:, 1:]


class RoIAlignAvg(RoIAlign):
    def forward(self, features, rois):
        assert rois.shape[1] == 5
        x = RoIAlignFunction(self.aligned_height + 1, self.aligned_width + 1, self.spatial_scale)(features, rois)
        a, b, c, d = _win2x2(x)  # avg_pool2d(kernel 2, stride 1) spelled out on views
        return (a + b + c + d) * 0.25


class RoIAlignMax(RoIAlign):
    def forward(self, features, rois):
        assert rois.shape[1] == 5
        x = RoIAlignFunction(self.aligned_height + 1, self.aligned_width + 1, self.spatial_scale)(features, rois)
        a, b, c, d = _win2x2(x)
        return torch.maximum(torch.maximum(a, b), torch.maximum(c, d))
