"""Evaluation path: utils.cal_mAP and the validate() loop against the reference's own
validate_single() / cal_mAP outputs (tests/golden/eval_256x512.npz and, at the size the metric is quoted on,
eval_512x1024.npz; made by tests/golden/make_golden_eval.py)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conftest import GOLDEN  # noqa: E402
sys.path.insert(0, GOLDEN)
import seeded_init as si  # noqa: E402
from test_host_functions import CFG  # noqa: E402

EVAL_SEEDS = dict(det=11, images=(41, 42), gts=(43, 44))
NAMES = ("frankfurt_000000_000294_leftImg8bit", "munster_000001_000019_leftImg8bit")


SIZES = [(256, 512), (512, 1024)]      # reduced size; BASELINE.json configs[1]'s


def _golden(size=(256, 512)):
    return np.load(os.path.join(GOLDEN, "eval_%dx%d.npz" % size))


def _lines(z, key):
    return str(z[key]).splitlines(True)


def eval_loader(H, W, G):
    imgs = [si.synth_images(s, H, W)[0] for s in EVAL_SEEDS['images']]
    gts = [si.synth_gts(G, s, H, W) for s in EVAL_SEEDS['gts']]
    info = torch.tensor([[H, W, 1.0]])
    return [(img, info.clone(), g.clone(), ["leftImg8bit/val/city/%s.png" % n]) for img, g, n in zip(imgs, gts, NAMES)]


def parse_rows(lines):
    rows = [l.split() for l in lines]
    return [(r[0], int(r[6]), np.array([float(v) for v in r[1:6]])) for r in rows]


def match_fraction(want, got, box_tol=0.05, score_tol=1e-4):
    """fraction of `want` rows with a (same image, same class) row of `got` within the tolerances"""
    hit = 0
    for n, c, v in want:
        cand = [g for (gn, gc, g) in got if gn == n and gc == c]
        if cand and min(max(np.abs(g[:4] - v[:4]).max() / box_tol, abs(g[4] - v[4]) / score_tol) for g in cand) <= 1.0:
            hit += 1
    return hit / max(len(want), 1)


@pytest.mark.parametrize("tag", ["", "3"])
def test_cal_map_matches_reference(tag):
    from scda_amd.dropin.utils import cal_mAP as C
    z = _golden()
    gl, sl = _lines(z, "meta" + tag), _lines(z, "synth_results" + tag)
    with np.errstate(all="ignore"):
        ap, mr = C.cal_mAP(C.parse_gts(gl, 9), C.parse_res(sl), 9, 0.5)
        m = C.Cal_MAP1(sl, gl, 9)
    assert np.array_equal(ap, z["ap_synth" + tag], equal_nan=True)          # same arithmetic, same order: bit-equal
    assert np.array_equal(mr, z["max_recall_synth" + tag], equal_nan=True)
    assert (np.isnan(m) and np.isnan(z["mAP_synth" + tag])) or m == float(z["mAP_synth" + tag])


def test_cal_map_edge_cases(tmp_path):
    from scda_amd.dropin.utils import cal_mAP as C
    z = _golden()
    gl = _lines(z, "meta")
    # a class without detections raises, exactly as the reference does on the random-init run
    assert bool(z["empty_class_raises"])
    with pytest.raises(ValueError), np.errstate(all="ignore"):
        C.cal_mAP(C.parse_gts(gl, 9), C.parse_res(_lines(z, "results")), 9, 0.5)
    # IoU conventions: +1 areas, strict overlap, first maximum
    assert C.calIoU([0, 0, 9, 9, 1.0, "a"], [[0, 0, 9, 9]]) == (1.0, 0)
    assert C.calIoU([0, 0, 9, 9, 1.0, "a"], [[9, 9, 20, 20]]) == (-1, -1)          # touching edge is not an overlap
    assert C.calIoU([0, 0, 9, 9, 1.0, "a"], []) == (-1, -1)
    assert C.calIoU([0, 0, 9, 9, 1.0, "a"], [[0, 0, 4, 9], [5, 0, 9, 9], [0, 0, 9, 4]])[1] == 0   # three equal IoUs
    assert C.parse_res(["img 1.9 2.1 -0.5 7.99 0.25 3\n"])[3] == [[1, 2, 0, 7, 0.25, "img"]]
    # Cal_MAP concatenates the per-rank files
    rows = _lines(z, "synth_results3")
    (tmp_path / "results.txt.rank0").write_text("".join(rows[::2]))
    (tmp_path / "results.txt.rank1").write_text("".join(rows[1::2]))
    (tmp_path / "meta.txt").write_text(str(z["meta3"]))
    with np.errstate(all="ignore"):
        m = C.Cal_MAP(str(tmp_path), str(tmp_path / "meta.txt"), 9)
    assert len((tmp_path / "results.txt").read_text().splitlines()) == len(rows)
    # splitting across ranks changes the order of tied scores only; the synthetic ties are duplicates -> same mAP
    assert abs(m - float(z["mAP_synth3"])) < 1e-12


@pytest.mark.parametrize("size", SIZES, ids=lambda s: "%d-%d" % s)
def test_validate_loop_with_oracle_detector_matches_reference(size, tmp_path):
    """the validate() loop driven by the CPU oracle detector reproduces the reference's validate_single() output"""
    from oracle import torch_ref as R
    from scda_amd.evaluate import validate
    z = _golden(size)
    H, W, G = int(z["H"]), int(z["W"]), int(z["G"])
    torch.manual_seed(1)
    det = R.build_models(CFG)[0]
    si.seeded_reinit(det, EVAL_SEEDS['det'], 'det')
    R.use_cpu_backend()
    try:
        rc = validate(eval_loader(H, W, G), det, CFG, str(tmp_path), score=False)
    finally:
        R.reset_backend()
    assert rc == float(z["recall"])
    got = (tmp_path / "results.txt.rank0").read_text().splitlines(True)
    want = _lines(z, "results")
    assert len(got) == len(want)
    assert [l.split()[0] + l.split()[6] for l in got] == [l.split()[0] + l.split()[6] for l in want]
    a = np.array([[float(v) for v in l.split()[1:6]] for l in got])
    b = np.array([[float(v) for v in l.split()[1:6]] for l in want])
    assert np.abs(a - b).max() < 2e-3, np.abs(a - b).max()


@pytest.mark.gpu
@pytest.mark.parametrize("size", SIZES, ids=lambda s: "%d-%d" % s)
def test_validate_on_device_matches_reference(size, cuda, tmp_path):
    """the HIP detector in eval mode through validate(): same recall, same detections as the reference's run"""
    import scda_amd.dropin as dropin
    dropin.install()
    from models.faster_rcnn import vgg_adver_expansion_cluster as V
    from scda_amd.evaluate import validate
    z = _golden(size)
    H, W, G = int(z["H"]), int(z["W"]), int(z["G"])
    torch.manual_seed(1)
    det = V.vgg16(pretrained=False, cfg=dict(CFG['shared'], gan_model_flag=2))
    si.seeded_reinit(det, EVAL_SEEDS['det'], 'det')
    det = det.to(cuda)
    rc = validate(eval_loader(H, W, G), det, CFG, str(tmp_path), score=False)
    assert abs(rc - float(z["recall"])) <= 1.0 / (2 * G) + 1e-9       # at most one ground truth differs in the recall count
    got = parse_rows((tmp_path / "results.txt.rank0").read_text().splitlines(True))
    want = parse_rows(_lines(z, "results"))
    assert len(got) == len(want)
    frac = match_fraction(want, got)
    assert frac >= 0.95, frac


@pytest.mark.gpu
def test_eval_forward_on_odd_sized_image(cuda):
    """200x312 input: the pooled feature maps are 100x156, 50x78, 25x39, 12x19 -- odd extents on the way (floor-mode pooling,
    ragged conv tiles, a 12x19 anchor grid).  Detections of the HIP detector vs the CPU oracle detector, same weights."""
    import scda_amd.dropin as dropin
    dropin.install()
    from models.faster_rcnn import vgg_adver_expansion_cluster as V
    from oracle import torch_ref as R
    H, W = 200, 312
    cfg = dict(CFG)
    torch.manual_seed(1)
    ref = R.build_models(CFG)[0]
    si.seeded_reinit(ref, EVAL_SEEDS['det'], 'det')
    det = V.vgg16(pretrained=False, cfg=dict(CFG['shared'], gan_model_flag=2))
    si.seeded_reinit(det, EVAL_SEEDS['det'], 'det')
    det = det.to(cuda).eval()
    ref.eval()
    img = si.synth_images(77, H, W)[0]
    x = {'cfg': cfg, 'image_info': torch.tensor([[H, W, 1.0]]), 'ground_truth_bboxes': None, 'ignore_regions': None}
    R.use_cpu_backend()
    try:
        with torch.no_grad():
            want = ref(dict(x, image=img))['predict']
    finally:
        R.reset_backend()
    with torch.no_grad():
        got = det(dict(x, image=img.to(cuda)))['predict']
    assert abs(got[0].shape[0] - want[0].shape[0]) <= 3                     # proposals after NMS
    g, w = got[1].cpu().numpy(), want[1].numpy()
    assert g.shape[1] == 7 and abs(g.shape[0] - w.shape[0]) <= 2
    hit = 0
    for row in w:
        d = np.abs(g[:, 1:5] - row[1:5]).max(axis=1) + 1e3 * (g[:, 6] != row[6]) + 1e3 * (np.abs(g[:, 5] - row[5]) > 1e-4)
        hit += d.min() < 0.05
    assert hit >= 0.95 * len(w), (hit, len(w))
