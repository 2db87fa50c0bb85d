"""API path of the reference (models/faster_rcnn/init.py); the initialisers live in scda_amd/dropin/_impl."""
from scda_amd.dropin._impl.initialisers import gaussian_weights_init, xavier_weights_init  # noqa: F401
