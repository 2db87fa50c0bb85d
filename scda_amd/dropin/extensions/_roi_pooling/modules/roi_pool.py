"""API path of the reference (extensions/_roi_pooling/modules/roi_pool.py); the module lives in scda_amd/dropin/_impl."""
from scda_amd.dropin._impl.roi_pool_module import _RoIPooling  # noqa: F401
