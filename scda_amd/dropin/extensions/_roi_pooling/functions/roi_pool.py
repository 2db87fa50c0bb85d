"""RoIPoolFunction(pooled_h, pooled_w, scale)(features, rois): the reference builds an old-style Function
instance and calls it (extensions/_roi_pooling/functions/roi_pool.py:6-42); this keeps that call shape on
top of a static autograd.Function."""
import torch

from scda_amd import native as N
from scda_amd.autograd_ops import RoIPoolFn


class RoIPoolFunction(object):
    def __init__(self, pooled_height, pooled_width, spatial_scale):
        self.pooled_height = int(pooled_height)
        self.pooled_width = int(pooled_width)
        self.spatial_scale = float(spatial_scale)

    def __call__(self, features, rois):
        if not (torch.is_grad_enabled() and features.requires_grad):
            # nothing will be differentiated through this call (target-domain branch, evaluation): no argmax output
            if not features.is_contiguous() or not rois.is_contiguous():
                raise AssertionError("RoIPool needs contiguous features and rois")
            return N.roi_pool_fwd(features, rois, self.pooled_height, self.pooled_width, self.spatial_scale, want_argmax=False)[0]
        return RoIPoolFn.apply(features, rois, self.pooled_height, self.pooled_width, self.spatial_scale)
