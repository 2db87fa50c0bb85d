"""bbox_overlaps(boxes f32[N,4+], query f32[K,4+]) -> f32[N,K]: extensions/_cython_bbox/cython_bbox.pyx:32-73."""
import numpy as np

from scda_amd.dropin import backend


def bbox_overlaps(boxes, query_boxes):
    if boxes.dtype != np.float32 or query_boxes.dtype != np.float32:
        raise ValueError("Buffer dtype mismatch, expected 'DTYPE_t' (float32)")  # what the Cython signature enforces
    return backend.bbox_overlaps(boxes, query_boxes)
