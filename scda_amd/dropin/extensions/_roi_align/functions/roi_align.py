"""RoIAlignFunction(aligned_h, aligned_w, scale)(features, rois) -- extensions/_roi_align/functions/roi_align.py:7-51.
Like the reference it exists on the accelerator only (the reference raises NotImplementedError on CPU, :30-31)."""
from scda_amd.autograd_ops import RoIAlignFn


class RoIAlignFunction(object):
    def __init__(self, aligned_height, aligned_width, spatial_scale):
        self.aligned_width = int(aligned_width)
        self.aligned_height = int(aligned_height)
        self.spatial_scale = float(spatial_scale)

    def __call__(self, features, rois):
        if not features.is_cuda:
            raise NotImplementedError
        return RoIAlignFn.apply(features, rois, self.aligned_height, self.aligned_width, self.spatial_scale)
