"""API path of the reference (functions/predict_bbox.py); the implementation lives in scda_amd/dropin/_impl."""
from scda_amd.dropin._impl.box_prediction import compute_predicted_bboxes  # noqa: F401
