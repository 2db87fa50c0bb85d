"""API path of the reference (extensions/_roi_align/modules/roi_align.py); the modules live in scda_amd/dropin/_impl."""
from scda_amd.dropin._impl.roi_align_modules import RoIAlign, RoIAlignAvg, RoIAlignMax  # noqa: F401
