"""Generates tests/golden/nms_ref.npz with the REFERENCE's own NMS: extensions/_cython_bbox/cython_nms.pyx compiled unmodified
by oracle/build_ref.py into oracle/_ref/cython_nms.cpython-39-*.so.

Run with the interpreter that built it (the only one that can import it):
    python oracle/build_ref.py && /opt/conda/bin/python3.9 tests/golden/make_golden_nms.py

The file holds outputs only (keep lists, soft-NMS results) plus a sha256 of every seeded input (tests/nms_cases.py regenerates the
inputs bit-identically under the system numpy and checks the digest).  cython_nms.nms (:37-87) is the greedy float32 IoU(+1)
suppression of nms_kernel.cu with ">=" instead of ">" at the threshold; nms_cases.make() guarantees that no pair of boxes sits
exactly on the threshold, so both give the same keep set, and with strictly decreasing scores np.where(suppressed == 0)[0]
(ascending original indices) IS the keep list in score order."""
import glob
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nms_cases  # noqa: E402


def load_ref():
    hits = glob.glob(os.path.join(ROOT, "oracle", "_ref", "cython_nms.cpython-39*.so"))
    assert hits, "run `python oracle/build_ref.py` first"
    if not hasattr(np, "int"):          # cython_nms.pyx:49 says np.zeros(..., dtype=np.int): a RUN-time alias numpy 1.24 dropped
        np.int = int
    spec = importlib.util.spec_from_file_location("cython_nms", hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    assert sys.version_info[:2] == (3, 9), "run with /opt/conda/bin/python3.9 (see the docstring)"
    ref = load_ref()
    out = {"numpy_version": np.array(np.__version__)}
    for name, n, kind, thresh in nms_cases.CASES:
        dets = nms_cases.make(name)
        assert not nms_cases.tie_pairs(dets, thresh)
        keep = ref.nms(dets, np.float32(thresh))
        out[name + "_sha256"] = np.array(nms_cases.digest(dets))
        out[name + "_keep"] = keep.astype(np.int32)
        print("%-22s n=%5d thresh=%.1f -> keeps %5d" % (name, n, thresh, len(keep)))
    # unsorted scores: the reference orders by scores.argsort()[::-1] itself (:45) and returns ascending ORIGINAL indices
    rs = np.random.RandomState(77)
    dets = nms_cases.make("clustered_2000_t05")
    perm = rs.permutation(len(dets))
    out["shuffled_perm"] = perm.astype(np.int32)
    out["shuffled_keep"] = ref.nms(np.ascontiguousarray(dets[perm]), np.float32(0.5)).astype(np.int32)
    # soft-NMS (:98-203), the three methods, on 300 clustered boxes (float32 in, (boxes, inds) out)
    small = nms_cases.make("rpn_300_t07")
    for method in (0, 1, 2):
        boxes, inds = ref.soft_nms(small.copy(), 0.5, 0.3, 0.001, method)
        out["soft_m%d_boxes" % method] = np.asarray(boxes, dtype=np.float32)
        out["soft_m%d_inds" % method] = np.asarray(inds, dtype=np.int32)
        print("soft_nms method %d -> %d boxes" % (method, len(inds)))
    np.savez_compressed(os.path.join(HERE, "nms_ref.npz"), **out)


if __name__ == "__main__":
    main()
