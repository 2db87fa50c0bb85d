"""Generates tests/golden/mask_targets_ref.npz with the REFERENCE's compute_mask_targets / generate_mask_labels
(functions/mask.py:51-179), imported UNMODIFIED through tests/golden/ref_harness.py.  Run in the build container:
    python tests/golden/make_golden_mask_targets.py

One stand-in: `cv2` is absent from this image, so the harness's cv2 stub gets `resize` = the drop-in's restatement of OpenCV's
8-bit INTER_LINEAR (scda_amd.dropin.functions.mask.resize_linear_u8).  The fixture therefore pins everything in the target
contract EXCEPT the resize arithmetic: RoI selection, integer clipping, IoU threshold, the np.random.choice draw, class planes,
-1 ignore labels, the all-ignore placeholder, output layout and dtypes."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402
import mask_cases as mcases  # noqa: E402


def main():
    ns = ref_harness.import_reference()
    from scda_amd.dropin.functions.mask import resize_linear_u8
    sys.modules["cv2"].resize = lambda img, size: resize_linear_u8(img, size[0], size[1])
    ns.mask.cv2 = sys.modules["cv2"]
    out = {}
    for name in mcases.CASES:
        props, gts, masks, info, cfg = mcases.make(name)
        np.random.seed(11)
        rois, labels = ns.mask.compute_mask_targets(torch.from_numpy(props), cfg, torch.from_numpy(gts), torch.from_numpy(masks),
                                                    torch.from_numpy(info))
        out[name + "_rois"] = rois.numpy()
        out[name + "_labels"] = labels.numpy().astype(np.int8)          # values in {-1, 0, 1}
        assert np.array_equal(out[name + "_labels"].astype(np.float32), labels.numpy())
        out[name + "_rng_after"] = np.array(np.random.randint(1 << 30))  # the RNG stream position after the call
        print(name, tuple(rois.shape), tuple(labels.shape))
    # predict_masks (functions/mask.py:21-49; PIL is present here, so this one runs entirely on the reference's own code)
    rois, heat, info = mcases.predict_case()
    pm = ns.mask.predict_masks(torch.from_numpy(rois), torch.from_numpy(heat), info)
    out["predict_masks"] = np.stack(pm).astype(np.float32)
    print("predict_masks", out["predict_masks"].shape)
    np.savez_compressed(os.path.join(HERE, "mask_targets_ref.npz"), **out)


if __name__ == "__main__":
    main()
