"""Generates tests/golden/roi_pool_ref.npz with the REFERENCE's own RoI max-pool: extensions/_roi_pooling/modules/roi_pool_py.py
(:7-47), imported UNMODIFIED through the legacy-semantics shims of tests/golden/ref_harness.py (identity .cuda(), torch<=0.3
`max` keeping the reduced dimension, 0.3-style row indexing).  Run in the build container:
    python tests/golden/make_golden_roipool.py

Holds outputs only: for the two small cases the whole [R,C,PH,PW] array, for the full-size case (12.8 M floats) a sha256 of its
bytes and every 251st element; plus a sha256 of each seeded input (tests/roipool_cases.py regenerates them bit-identically).
roi_pool_py.py returns values, not argmax indices; the tests derive the argmax checks from the values (features[argmax] == value,
no earlier element of the bin holds it)."""
import hashlib
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402
import roipool_cases as rc  # noqa: E402

STRIDE = 251


def main():
    torch.set_num_threads(1)                      # thousands of tiny reductions: the pool only adds latency
    run = ref_harness.import_reference_roi_pool_py()
    out = {"stride": np.array(STRIDE)}
    for name, shape, R, PH, PW, scale in rc.CASES:
        feat, rois = rc.make(name)
        rc.check_clean(name, rois)
        t = time.time()
        ref = run(torch.from_numpy(feat), torch.from_numpy(rois), PH, PW, scale).numpy()
        assert ref.dtype == np.float32 and ref.shape == (R, shape[1], PH, PW)
        out[name + "_in_sha256"] = np.array(rc.digest(feat, rois))
        out[name + "_out_sha256"] = np.array(hashlib.sha256(ref.tobytes()).hexdigest())
        if ref.size <= 1 << 16:
            out[name + "_out"] = ref
        else:
            out[name + "_out_sample"] = ref.reshape(-1)[::STRIDE].copy()
        empty = int((ref.reshape(R, -1) == 0).all(1).sum())
        print("%-26s feat %s  R=%d  %dx%d  -> %.1f s, %d RoIs entirely empty" % (name, shape, R, PH, PW, time.time() - t, empty))
    np.savez_compressed(os.path.join(HERE, "roi_pool_ref.npz"), **out)


if __name__ == "__main__":
    main()
