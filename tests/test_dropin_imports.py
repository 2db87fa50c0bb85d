"""The drop-in claim, checked where the reference tree exists (this build container): with scda_amd/dropin FIRST on
sys.path the reference's own driver module `tools/faster_rcnn_train_val.py` imports -- its whole import block (:18-58)
resolves -- the replaced symbols come from this repository, everything else from the reference checkout, and every replaced
callable accepts the reference's positional arguments (signatures compared against the reference sources by `ast`, because
the reference's ffi-era modules cannot be imported on torch 2.x).  Runs in a subprocess: the import installs stubs for the
third-party packages this image lacks (cv2, torchvision, the compiled pycocotools mask module) and sets the
multiprocessing start method.  CPU only, skipped where /root/reference is absent (the GPU box)."""
import ast
import json
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

REF = os.environ.get("SCDA_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")

PROBE = textwrap.dedent(r'''
    import inspect, json, os, sys, types
    root, ref = sys.argv[1], sys.argv[2]
    sys.path[:0] = [os.path.join(root, "scda_amd", "dropin"), root, ref]
    # third-party packages the reference driver imports and this image does not have
    cv2 = types.ModuleType("cv2"); cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda *_: None); cv2.setNumThreads = lambda *_: None
    sys.modules["cv2"] = cv2
    tv = types.ModuleType("torchvision"); tvt = types.ModuleType("torchvision.transforms")
    for n in ("Normalize", "ToTensor", "Compose"):
        setattr(tvt, n, type(n, (), {"__init__": lambda self, *a, **k: None, "__call__": lambda self, x: x}))
    tv.transforms = tvt; sys.modules["torchvision"] = tv; sys.modules["torchvision.transforms"] = tvt
    mk = types.ModuleType("datasets.pycocotools._mask")
    for a in ("iou", "merge", "frPyObjects", "encode", "decode", "area", "toBbox"):
        setattr(mk, a, lambda *x, **k: None)
    sys.modules["datasets.pycocotools._mask"] = mk
    import numpy as np
    np.float, np.int = float, int

    import tools.faster_rcnn_train_val as T          # the reference's driver, unmodified
    import extensions
    from extensions import nms, RoIPool
    from extensions._roi_align.modules.roi_align import RoIAlign, RoIAlignAvg, RoIAlignMax
    from extensions._roi_align.functions.roi_align import RoIAlignFunction
    from extensions._roi_pooling.functions.roi_pool import RoIPoolFunction
    from extensions._focal_loss.focal_loss import SigmoidFocalLossFunction, SoftmaxFocalLossFunction
    from extensions._bbox_helper.bbox_helper import overlap
    from extensions._cython_bbox import cython_bbox, cython_nms
    import models.head, models.losses, functions.anchor_target, functions.rpn_proposal, functions.proposal_target
    import functions.predict_bbox, functions.mask, utils.anchor_helper, utils.bbox_helper, utils.distributed_utils
    import utils.lr_helper, utils.load_helper, utils.log_helper

    def where(obj):
        return os.path.realpath(inspect.getsourcefile(obj))

    def params(fn):
        return [p.name for p in inspect.signature(fn).parameters.values()
                if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD) and p.name != "self"]

    out = {"files": {}, "sigs": {}}
    for name, obj in {
        "driver": T, "vgg16": T.vgg16_FasterRCNN, "vgg16_bn": T.vgg16bn_FasterRCNN, "GAN_dis_AE": T.GAN_dis_AE,
        "GAN_dis_AE_patch": T.GAN_dis_AE_patch, "GAN_decoder_AE": T.GAN_decoder_AE, "GAN_decoder_AE_32": T.GAN_decoder_AE_32,
        "dist_init": T.dist_init, "average_gradients": T.average_gradients, "broadcast_params": T.broadcast_params,
        "Cal_MAP": T.Cal_MAP, "bbox_helper": T.bbox_helper, "IterExponentialLR": T.IterExponentialLR,
        "restore_from": T.restore_from, "init_log": T.init_log, "ExampleDataset": T.ExampleDataset,
        "nms": nms, "RoIPool": RoIPool, "RoIAlignAvg": RoIAlignAvg, "SigmoidFocalLossFunction": SigmoidFocalLossFunction,
        "overlap": overlap, "cython_bbox": cython_bbox.bbox_overlaps, "cython_nms": cython_nms.nms,
        "NaiveRpnHead": models.head.NaiveRpnHead, "compute_anchor_targets": functions.anchor_target.compute_anchor_targets,
        "compute_rpn_proposals": functions.rpn_proposal.compute_rpn_proposals,
        "compute_proposal_targets": functions.proposal_target.compute_proposal_targets,
        "compute_predicted_bboxes": functions.predict_bbox.compute_predicted_bboxes,
        "compute_cluster_targets": functions.mask.compute_cluster_targets,
    }.items():
        out["files"][name] = where(obj)
    for key, fn in {
        "extensions/_nms/pth_nms.py:pth_nms": nms,
        "extensions/_roi_pooling/modules/roi_pool.py:_RoIPooling.__init__": RoIPool.__init__,
        "extensions/_roi_pooling/modules/roi_pool.py:_RoIPooling.forward": RoIPool.forward,
        "extensions/_roi_pooling/functions/roi_pool.py:RoIPoolFunction.__init__": RoIPoolFunction.__init__,
        "extensions/_roi_align/modules/roi_align.py:RoIAlign.__init__": RoIAlign.__init__,
        "extensions/_roi_align/modules/roi_align.py:RoIAlign.forward": RoIAlign.forward,
        "extensions/_roi_align/modules/roi_align.py:RoIAlignAvg.__init__": RoIAlignAvg.__init__,
        "extensions/_roi_align/modules/roi_align.py:RoIAlignMax.forward": RoIAlignMax.forward,
        "extensions/_roi_align/functions/roi_align.py:RoIAlignFunction.__init__": RoIAlignFunction.__init__,
        "extensions/_focal_loss/focal_loss.py:SigmoidFocalLossFunction.__init__": SigmoidFocalLossFunction.__init__,
        "extensions/_focal_loss/focal_loss.py:SigmoidFocalLossFunction.forward": SigmoidFocalLossFunction.__call__,
        "extensions/_focal_loss/focal_loss.py:SoftmaxFocalLossFunction.__init__": SoftmaxFocalLossFunction.__init__,
        "extensions/_focal_loss/focal_loss.py:SoftmaxFocalLossFunction.forward": SoftmaxFocalLossFunction.__call__,
        "extensions/_bbox_helper/bbox_helper.py:overlap": overlap,
        "models/head.py:NaiveRpnHead.__init__": models.head.NaiveRpnHead.__init__,
        "models/head.py:NaiveRpnHead.forward": models.head.NaiveRpnHead.forward,
        "models/faster_rcnn/vgg_adver_expansion_cluster.py:vgg16": T.vgg16_FasterRCNN,
        "models/faster_rcnn/vgg_adver_expansion_cluster.py:vgg16_bn": T.vgg16bn_FasterRCNN,
        "models/faster_rcnn/faster_rcnn_adver_expansion_reweight_cluster.py:GAN_dis_AE.__init__": T.GAN_dis_AE.__init__,
        "models/faster_rcnn/faster_rcnn_adver_expansion_reweight_cluster.py:GAN_dis_AE.forward": T.GAN_dis_AE.forward,
        "models/faster_rcnn/faster_rcnn_adver_expansion_reweight_cluster.py:GAN_dis_AE_patch.__init__": T.GAN_dis_AE_patch.__init__,
        "models/faster_rcnn/faster_rcnn_adver_expansion_reweight_cluster.py:GAN_dis_AE_patch.forward": T.GAN_dis_AE_patch.forward,
        "models/faster_rcnn/faster_rcnn_adver_expansion_reweight_cluster.py:GAN_decoder_AE.__init__": T.GAN_decoder_AE.__init__,
        "models/faster_rcnn/faster_rcnn_adver_expansion_reweight_cluster.py:GAN_decoder_AE.forward": T.GAN_decoder_AE.forward,
        "models/faster_rcnn/faster_rcnn_adver_expansion_reweight_cluster.py:FasterRCNN_AdEx.forward":
            sys.modules["models.faster_rcnn.faster_rcnn_adver_expansion_reweight_cluster"].FasterRCNN_AdEx.forward,
        "utils/distributed_utils.py:dist_init": T.dist_init,
        "utils/distributed_utils.py:average_gradients": T.average_gradients,
        "utils/distributed_utils.py:broadcast_params": T.broadcast_params,
        "utils/cal_mAP.py:Cal_MAP": T.Cal_MAP,
        "functions/anchor_target.py:compute_anchor_targets": functions.anchor_target.compute_anchor_targets,
        "functions/rpn_proposal.py:compute_rpn_proposals": functions.rpn_proposal.compute_rpn_proposals,
        "functions/proposal_target.py:compute_proposal_targets": functions.proposal_target.compute_proposal_targets,
        "functions/predict_bbox.py:compute_predicted_bboxes": functions.predict_bbox.compute_predicted_bboxes,
        "functions/mask.py:compute_cluster_targets": functions.mask.compute_cluster_targets,
        "utils/anchor_helper.py:get_anchors_over_plane": utils.anchor_helper.get_anchors_over_plane,
        "utils/bbox_helper.py:compute_loc_bboxes": utils.bbox_helper.compute_loc_bboxes,
        "utils/bbox_helper.py:clip_bbox": utils.bbox_helper.clip_bbox,
    }.items():
        out["sigs"][key] = params(fn)
    print("PROBE" + json.dumps(out))
''')


def reference_params(path, qual):
    """positional parameter names (without self) of function / Class.method `qual` in reference file `path`, via ast"""
    tree = ast.parse(open(os.path.join(REF, path)).read())
    parts = qual.split(".")
    body = tree.body
    node = None
    for i, part in enumerate(parts):
        node = next(n for n in body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name == part)
        body = getattr(node, "body", [])
    names = [a.arg for a in node.args.args]
    return names[1:] if names and names[0] in ("self", "ctx") else names   # roi_pool.py:7 calls its `self` "ctx"


@pytest.fixture(scope="module")
def probe():
    r = subprocess.run([sys.executable, "-c", PROBE, ROOT, REF], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONPATH=""))
    line = [l for l in r.stdout.splitlines() if l.startswith("PROBE")]
    assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads(line[0][5:])


def test_reference_driver_imports_with_dropin_first(probe):
    mine = os.path.realpath(os.path.join(ROOT, "scda_amd")) + os.sep
    ref = os.path.realpath(REF) + os.sep
    from_here = ("vgg16", "vgg16_bn", "GAN_dis_AE", "GAN_dis_AE_patch", "GAN_decoder_AE", "GAN_decoder_AE_32", "dist_init",
                 "average_gradients", "broadcast_params", "Cal_MAP", "bbox_helper", "nms", "RoIPool", "RoIAlignAvg",
                 "SigmoidFocalLossFunction", "overlap", "cython_bbox", "cython_nms", "NaiveRpnHead", "compute_anchor_targets",
                 "compute_rpn_proposals", "compute_proposal_targets", "compute_predicted_bboxes", "compute_cluster_targets")
    for k in from_here:
        assert probe["files"][k].startswith(mine), (k, probe["files"][k])
    for k in ("driver", "IterExponentialLR", "restore_from", "init_log", "ExampleDataset"):   # untouched reference modules
        assert probe["files"][k].startswith(ref), (k, probe["files"][k])


def test_replaced_callables_accept_the_reference_arguments(probe):
    """every positional parameter of the reference's definition exists, in the same position, in the replacement (the
    replacement may append optional ones, e.g. async_op / max_keep)"""
    bad = {}
    for key, got in probe["sigs"].items():
        path, qual = key.split(":")
        want = reference_params(path, qual)
        if got[:len(want)] != want:
            bad[key] = (want, got)
    assert not bad, bad
