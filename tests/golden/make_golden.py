#!/usr/bin/env python
"""Generate the committed golden vectors by running the REFERENCE's own Python
(/root/reference, imported unmodified through ref_harness.py) on seeded inputs.

Run in the build container only:   python tests/golden/make_golden.py [--only l2 ...]
Outputs: tests/golden/*.npz  (data only: inputs/seeds and the reference's outputs).

Fixtures
  bbox_overlaps.npz   reference Cython bbox_overlaps (built unmodified) on seeded boxes
  l2_G{3,12,30}.npz   anchors, anchor targets, RPN proposals, proposal targets,
                      k-means cluster targets, crop corners at config-2 size
                      (feature 32x64, A=15, 30720 anchors, 512 RoIs, 4 clusters)
  predict_bbox.npz    test-time box prediction (per-class NMS, top-100)
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

CFG_PATH = os.path.join(rh.REF, "examples/faster-rcnn/cityscapes/vgg/config_512.json")


def load_cfg():
    cfg = json.load(open(CFG_PATH))
    for k in cfg:
        if k != "shared":
            cfg[k].update(cfg["shared"])
    return cfg


def synth_gts(G, seed, H=512, W=1024):
    """Synthetic ground truth, SURVEY.md 8(d): log-uniform sizes, integer corners, classes 1..8."""
    r = np.random.RandomState(seed)
    w = np.exp(r.uniform(np.log(16), np.log(400), G))
    h = np.exp(r.uniform(np.log(16), np.log(300), G))
    x1 = r.uniform(0, W - 1 - w)
    y1 = r.uniform(0, H - 1 - h)
    box = np.stack([np.floor(x1), np.floor(y1), np.ceil(x1 + w), np.ceil(y1 + h)], 1)
    box[:, 2] = np.minimum(box[:, 2], W - 1)
    box[:, 3] = np.minimum(box[:, 3], H - 1)
    cls = r.randint(1, 9, G)
    return np.concatenate([box, cls[:, None]], 1).astype(np.float32)[None]  # [1,G,5]


def synth_rpn_outputs(seed, A=15, fh=32, fw=64):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(1, fh, fw, A, 2, generator=g) * 2.0
    prob = torch.softmax(logits, -1)  # [1,fh,fw,A,2]
    cls = prob.reshape(1, fh, fw, A * 2).permute(0, 3, 1, 2).contiguous()
    loc = (torch.randn(1, A * 4, fh, fw, generator=g) * 0.3).contiguous()
    return cls, loc


def gen_bbox_overlaps(ns, out):
    r = np.random.RandomState(7)
    cases = {}
    for name, (N, K) in {"small": (37, 5), "anchors": (3000, 12), "degenerate": (64, 8)}.items():
        b = r.uniform(0, 600, (N, 2)).astype(np.float32)
        wh = r.uniform(0.5, 300, (N, 2)).astype(np.float32)
        boxes = np.concatenate([b, b + wh], 1).astype(np.float32)
        q = r.uniform(0, 600, (K, 2)).astype(np.float32)
        qwh = r.uniform(1, 300, (K, 2)).astype(np.float32)
        query = np.concatenate([q, q + qwh], 1).astype(np.float32)
        if name == "degenerate":
            boxes[:8] = query  # exact matches (IoU 1)
            boxes[8:16, 2:] = boxes[8:16, :2]  # zero-area boxes
            boxes[16:20] = np.round(boxes[16:20])  # integer corners
        ov = ns.cython_bbox.bbox_overlaps(boxes, query)
        cases[name + "_boxes"] = boxes
        cases[name + "_query"] = query
        cases[name + "_out"] = ov
    np.savez_compressed(os.path.join(out, "bbox_overlaps.npz"), **cases)
    print("bbox_overlaps.npz written")


def gen_l2(ns, out):
    cfg = load_cfg()
    image_info = np.array([[512, 1024, 1.0]], dtype=np.float32)
    anchors = ns.anchor_helper.get_anchors_over_plane(32, 64, cfg["shared"]["anchor_ratios"],
                                                      cfg["shared"]["anchor_scales"], cfg["shared"]["anchor_stride"])
    for G in (3, 12, 30):
        seed = 100 + G
        gts = synth_gts(G, seed)
        d = {"gts": gts, "image_info": image_info, "seed": np.int64(seed)}
        if G == 3:
            d["anchors"] = anchors  # float64 [30720,4]

        # --- a5 anchor targets (functions/anchor_target.py:16-116) ---
        np.random.seed(seed)
        cls_t, loc_t, loc_m, norm = ns.anchor_target.compute_anchor_targets(
            (1, 60, 32, 64), cfg["train_anchor_target_cfg"], torch.from_numpy(gts), torch.from_numpy(image_info), None)
        d["at_cls_targets"] = cls_t.numpy().astype(np.int8)  # [1,15,32,64]
        lt = loc_t.numpy()
        nz = np.nonzero(loc_m.numpy().reshape(-1))[0]
        d["at_loc_nz_index"] = nz.astype(np.int32)
        d["at_loc_targets_nz"] = lt.reshape(-1)[nz]
        d["at_loc_masks_sum"] = np.float64(loc_m.numpy().sum())
        d["at_normalizer"] = np.int64(norm)

        # --- a6 RPN proposals (functions/rpn_proposal.py:17-74) ---
        cls, loc = synth_rpn_outputs(seed)
        props = ns.rpn_proposal.compute_rpn_proposals(cls, loc, cfg["train_rpn_proposal_cfg"], image_info)
        d["rpn_seed"] = np.int64(seed)
        d["proposals"] = props.numpy()  # [<=2000,6]
        props_test = ns.rpn_proposal.compute_rpn_proposals(cls, loc, cfg["test_rpn_proposal_cfg"], image_info)
        d["proposals_test"] = props_test.numpy()  # [<=300,6]

        # --- a7 proposal targets (functions/proposal_target.py:17-177) ---
        np.random.seed(seed + 1)
        rois, labels, ploc_t, ploc_w = ns.proposal_target.compute_proposal_targets(
            props, cfg["train_proposal_target_cfg"], torch.from_numpy(gts), torch.from_numpy(image_info), None)
        d["pt_rois"] = rois.numpy()
        d["pt_labels"] = labels.numpy().astype(np.int16)
        pw = ploc_w.numpy()
        nzp = np.nonzero(pw.reshape(-1))[0]
        d["pt_loc_nz_index"] = nzp.astype(np.int32)
        d["pt_loc_targets_nz"] = ploc_t.numpy().reshape(-1)[nzp]

        # --- a8 cluster targets (functions/mask.py:183-237) ---
        # features row i = i, so the returned [4,128,4096] tensor spells out the chosen RoI indices
        feats = torch.arange(512, dtype=torch.float32)[:, None].repeat(1, 8)
        np.random.seed(seed + 2)
        cf, centres = ns.mask.compute_cluster_targets(rois, feats, N_cluster=4, threshold=128)
        d["ct_index"] = cf.numpy()[:, :, 0].astype(np.int16)  # [4,128]
        d["ct_centres"] = np.asarray(centres, dtype=np.float64)  # [4,2]
        # second cluster call on the target-style proposals (first 512 of the RPN output)
        pg = props[0:512, :5].contiguous()
        np.random.seed(seed + 3)
        cf2, centres2 = ns.mask.compute_cluster_targets(pg, feats[: pg.shape[0]], N_cluster=4, threshold=128)
        d["ct2_index"] = cf2.numpy()[:, :, 0].astype(np.int16)
        d["ct2_centres"] = np.asarray(centres2, dtype=np.float64)
        np.savez_compressed(os.path.join(out, f"l2_G{G}.npz"), **d)
        print(f"l2_G{G}.npz written: props {tuple(props.shape)} pos_anchors {int((cls_t == 1).sum())} "
              f"fg_rois {int((labels > 0).sum())}")


def gen_predict_bbox(ns, out):
    """Test-time path: functions/predict_bbox.py:13-66 (per-class NMS 0.5, top 100)."""
    cfg = load_cfg()
    image_info = np.array([[512, 1024, 1.0]], dtype=np.float32)
    cls, loc = synth_rpn_outputs(555)
    props = ns.rpn_proposal.compute_rpn_proposals(cls, loc, cfg["test_rpn_proposal_cfg"], image_info)
    rois = props[:, :5].contiguous()
    g = torch.Generator().manual_seed(556)
    R = rois.shape[0]
    pred_cls = torch.softmax(torch.randn(R, 9, generator=g) * 2, 1)
    pred_loc = torch.randn(R, 36, generator=g) * 0.5
    bb = ns.predict_bbox.compute_predicted_bboxes(rois, pred_cls, pred_loc, image_info, cfg["test_predict_bbox_cfg"])
    np.savez_compressed(os.path.join(out, "predict_bbox.npz"), rois=rois.numpy(), pred_cls=pred_cls.numpy(),
                        pred_loc=pred_loc.numpy(), image_info=image_info, bboxes=bb.numpy())
    print("predict_bbox.npz written", tuple(bb.shape))


def gen_corners(ns_driver, out):
    T = ns_driver.T
    r = np.random.RandomState(3)
    centres = np.concatenate([r.uniform(0, 1024, (64, 1)), r.uniform(0, 512, (64, 1))], 1)
    centres[:6] = [[0, 0], [1023.9, 511.9], [128, 128], [127.5, 384.5], [896, 384], [896.5, 100]]
    corners = np.array(T.get_corner_from_center(centres), dtype=np.int32)
    np.savez_compressed(os.path.join(out, "corners.npz"), centres=centres, corners=corners)
    print("corners.npz written")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    want = lambda k: a.only is None or k in a.only  # noqa: E731
    need_driver = want("corners") or want("train") or want("eval") or want("eval512")
    if need_driver:
        ns = rh.import_reference_driver(["--config", CFG_PATH, "--port", "1", "--dataset", "cityscapes", "--datadir", "x",
                                         "--arch", "vgg16_FasterRCNN", "--dist", "0", "--cluster_num", "4",
                                         "--threshold", "128", "--recon_size", "256"])
    else:
        ns = rh.import_reference()
    if want("bbox"):
        gen_bbox_overlaps(ns, HERE)
    if want("l2"):
        gen_l2(ns, HERE)
    if want("predict"):
        gen_predict_bbox(ns, HERE)
    if want("corners"):
        gen_corners(ns, HERE)
    if want("train"):
        import make_golden_model
        make_golden_model.generate(ns, HERE)
    if want("eval"):
        import make_golden_eval
        make_golden_eval.generate(ns, HERE)
    if want("eval512"):       # the same at the size the metric is quoted on (BASELINE.json configs[1]): 512 x 1024, 12 gt boxes per image
        import make_golden_eval
        make_golden_eval.generate(ns, HERE, H=512, W=1024, G=12)
    if want("data"):
        import make_golden_data
        make_golden_data.generate(ns, HERE)
