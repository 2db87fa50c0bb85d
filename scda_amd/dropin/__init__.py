"""Host-side mirror of the reference's operator / module interface.

This directory contains top-level packages with the reference's own names --
`extensions`, `models`, `functions`, `utils` -- so that
    PYTHONPATH=<repo>:<repo>/scda_amd/dropin:<reference checkout>
makes the reference's `tools/faster_rcnn_train_val.py` import THESE modules in
place of its CUDA/cffi ones.  Inside this repo the same
modules are imported as `scda_amd.dropin.<name>`.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def install():
    """Put the mirror packages first on sys.path (idempotent)."""
    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    root = os.path.dirname(os.path.dirname(HERE))
    if root not in sys.path:
        sys.path.insert(1, root)
