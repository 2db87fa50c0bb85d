"""RoIPoolFunction(pooled_h, pooled_w, scale)(features, rois): the reference builds an old-style Function
instance and calls it (extensions/_roi_pooling/functions/roi_pool.py:6-42); this keeps that call shape on
top of a static autograd.Function."""
from scda_amd.autograd_ops import RoIPoolFn


class RoIPoolFunction(object):
    def __init__(self, pooled_height, pooled_width, spatial_scale):
        self.pooled_height = int(pooled_height)
        self.pooled_width = int(pooled_width)
        self.spatial_scale = float(spatial_scale)

    def __call__(self, features, rois):
        return RoIPoolFn.apply(features, rois, self.pooled_height, self.pooled_width, self.spatial_scale)
