"""_RoIPooling(pooled_height, pooled_width, spatial_scale): nn.Module of extensions/_roi_pooling/modules/roi_pool.py:5-15."""
from torch.nn.modules.module import Module

from scda_amd.dropin.extensions._roi_pooling.functions.roi_pool import RoIPoolFunction


class _RoIPooling(Module):
    def __init__(self, pooled_height, pooled_width, spatial_scale):
        super().__init__()
        self.pooled_width = int(pooled_width)
        self.pooled_height = int(pooled_height)
        self.spatial_scale = float(spatial_scale)

    def forward(self, features, rois):
        assert rois.shape[1] == 5
        return RoIPoolFunction(self.pooled_height, self.pooled_width, self.spatial_scale)(features, rois)
