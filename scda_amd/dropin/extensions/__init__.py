"""`from extensions import nms, RoIPool` -- same surface as the reference's extensions/__init__.py:1-3."""
from scda_amd.dropin.extensions._nms.pth_nms import pth_nms as nms
from scda_amd.dropin.extensions._roi_pooling.modules.roi_pool import _RoIPooling as RoIPool

__all__ = ["nms", "RoIPool"]
