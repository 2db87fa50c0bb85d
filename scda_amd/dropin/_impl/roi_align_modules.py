"""RoIAlign / RoIAlignAvg / RoIAlignMax with the constructor arguments and attributes of the reference's
extensions/_roi_align/modules/roi_align.py:6-44.

RoIAlign samples an aligned_height x aligned_width grid per RoI (HIP kernel).  The Avg / Max variants sample one extra row
and column and then reduce every 2x2 neighbourhood (stride 1) -- written out on four shifted views instead of a pooling
call, so it is plain tensor arithmetic on the device and differentiable through the RoIAlign function."""
import torch
from torch.nn.modules.module import Module

from scda_amd.dropin.extensions._roi_align.functions.roi_align import RoIAlignFunction


class RoIAlign(Module):
    extra = 0            # additional samples per axis before the 2x2 reduction
    reduce2x2 = None     # callable(a, b, c, d) over the four shifted views, or None

    def __init__(self, aligned_height, aligned_width, spatial_scale):
        super().__init__()
        self.aligned_height, self.aligned_width = int(aligned_height), int(aligned_width)
        self.spatial_scale = float(spatial_scale)

    def forward(self, features, rois):
        kind = type(self)
        if kind.reduce2x2 is not None:
            assert rois.shape[1] == 5
        fn = RoIAlignFunction(self.aligned_height + kind.extra, self.aligned_width + kind.extra, self.spatial_scale)
        x = fn(features, rois)
        if kind.reduce2x2 is None:
            return x
        return kind.reduce2x2(x[:, :, :-1, :-1], x[:, :, :-1, 1:], x[:, :, 1:, :-1], x[:, :, 1:, 1:])


class RoIAlignAvg(RoIAlign):
    extra = 1
    channel_major = False   # True: [C, R, h, w] instead of [R, C, h, w] (set by a detector whose RoI head runs channel-major)

    def forward(self, features, rois):
        from scda_amd.autograd_ops import Avg2x2S1Fn, RoIAlignFn
        assert rois.shape[1] == 5
        if self.channel_major:
            x = RoIAlignFn.apply(features, rois, self.aligned_height + 1, self.aligned_width + 1, self.spatial_scale, True)
        else:
            x = RoIAlignFunction(self.aligned_height + 1, self.aligned_width + 1, self.spatial_scale)(features, rois)
        return Avg2x2S1Fn.apply(x)      # avg_pool2d(kernel_size=2, stride=1) as one kernel (forward and backward); per plane


class RoIAlignMax(RoIAlign):
    extra = 1
    reduce2x2 = staticmethod(lambda a, b, c, d: torch.maximum(torch.maximum(a, b), torch.maximum(c, d)))
