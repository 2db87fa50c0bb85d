"""mAP parity -- the north_star's acceptance line "mAP within +-0.3 of reference" (BASELINE.md section 1), on a detector that
actually detects.

A VGG16 Faster-R-CNN (+ the SCDA nets, the whole 4-phase iteration) is trained ON THE DEVICE on a handful of synthetic 256x512
images (coloured rectangles, colour = class) until its mAP@0.5 on them is far from trivial; the SAME checkpoint is then evaluated
  (a) by the CPU oracle detector (oracle/torch_ref.py RefDetector: torch-CPU convolutions, the C restatements of NMS / RoIPool /
      IoU -- the oracle that reproduces the reference's own train() and validate_single() outputs, tests/test_oracle_model.py,
      tests/test_eval_path.py), and
  (b) by the HIP detector,
both through the same validate() loop (tools/faster_rcnn_train_val.py:773-884) and utils.cal_mAP (:95-171).  Asserted: both
mAPs non-trivial, |mAP_hip - mAP_oracle| <= 0.3 points, equal numbers of result rows; and, on an UNSATURATED checkpoint evaluated on
held-out images, the result rows themselves (test_map_rows_of_an_unsaturated_checkpoint_on_held_out_images).

Run by hand to see the numbers / tune:  python tests/test_map_parity_gpu.py [iterations] [lr]"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import copy  # noqa: E402
from test_host_functions import CFG as _CFG  # noqa: E402

CFG = copy.deepcopy(_CFG)
for _k in CFG:
    CFG[_k]['gan_model_flag'] = 2

H, W, N_IMG, PER_IMG = 256, 512, 8, 3
SCALE = 1          # rectangle sizes scale with the image (set_size)


def set_size(h, w):
    """the image size of everything below: 256 x 512 (default) or the metric's own 512 x 1024 (rectangles twice as large)"""
    global H, W, SCALE
    H, W, SCALE = h, w, h // 256
PALETTE = np.array([[a, b, c] for a in (-0.9, 0.9) for b in (-0.9, 0.9) for c in (-0.9, 0.9)], np.float32)   # class k+1 -> colour k


def make_dataset(seed=7):
    """N_IMG images [1,3,H,W] in [-1,1]: faint noise + PER_IMG non-overlapping filled rectangles whose colour IS the class (1..8);
    every class occurs three times.  -> images, gts [1,PER_IMG,5] (x1,y1,x2,y2,class; integer corners), names"""
    rs = np.random.RandomState(seed)
    classes = np.concatenate([rs.permutation(8) + 1 for _ in range(N_IMG * PER_IMG // 8)])
    images, gts, names = [], [], []
    for i in range(N_IMG):
        img = (0.05 * rs.standard_normal((3, H, W))).astype(np.float32)
        boxes = []
        while len(boxes) < PER_IMG:
            w, h = SCALE * rs.randint(56, 150), SCALE * rs.randint(48, 110)
            x1, y1 = rs.randint(4, W - w - 4), rs.randint(4, H - h - 4)
            if any(not (x1 > b[2] + 8 or x1 + w < b[0] - 8 or y1 > b[3] + 8 or y1 + h < b[1] - 8) for b in boxes):
                continue
            c = int(classes[i * PER_IMG + len(boxes)])
            img[:, y1:y1 + h + 1, x1:x1 + w + 1] = PALETTE[c - 1][:, None, None] + 0.05 * rs.standard_normal((3, h + 1, w + 1))
            boxes.append([x1, y1, x1 + w, y1 + h, c])
        images.append(torch.from_numpy(np.clip(img, -1, 1))[None])
        gts.append(torch.tensor(boxes, dtype=torch.float32)[None])
        names.append("synth_%06d_leftImg8bit" % i)
    return images, gts, names


def meta_lines(names, gts):
    """the val meta list parse_gts() reads (utils/cal_mAP.py:16-47)"""
    out = []
    for i, (n, g) in enumerate(zip(names, gts)):
        g = g[0].numpy()
        out += ["# %d\n" % i, "val/city/%s.png\n" % n, "3\n", "%d\n" % H, "%d\n" % W, "0\n", "0\n", "%d\n" % len(g)]
        out += ["%d %d %d %d %d\n" % (b[4], b[0], b[1], b[2], b[3]) for b in g]
    return out


def loader(images, gts, names, device=None):
    info = torch.tensor([[H, W, 1.0]])
    return [(img if device is None else img.to(device), info.clone(), g.clone(), ["leftImg8bit/val/city/%s.png" % n])
            for img, g, n in zip(images, gts, names)]


def train_on_device(cuda, images, gts, iters, lr):
    from scda_amd.train_step import ScdaTrainer
    torch.manual_seed(3)
    np.random.seed(3)
    tr = ScdaTrainer(CFG, cuda, lr=lr, new_w=W, new_h=H)
    info = torch.tensor([[H, W, 1.0]])
    dev_imgs = [im.to(cuda) for im in images]
    hist = []
    for it in range(iters):
        k = it % N_IMG
        out = tr.step(dev_imgs[k], gts[k], info, dev_imgs[(k + 3) % N_IMG])     # another image of the set plays the target domain
        if it % 50 == 49 or it == iters - 1:
            hist.append((it + 1, float(out['rpn_cls']), float(out['rpn_loc']), float(out['rcnn_cls']), float(out['rcnn_loc']),
                         float(out['rcnn_acc'])))
    return tr, hist


def score(results_dir, meta_file):
    from scda_amd.dropin.utils import cal_mAP as C
    import contextlib
    import io
    with np.errstate(all="ignore"), contextlib.redirect_stdout(io.StringIO()):
        m = C.Cal_MAP(results_dir, meta_file, 9)
    rows = open(os.path.join(results_dir, "results.txt")).read().splitlines()
    return 100.0 * float(m), rows


def compare_rows(rows_a, rows_b, box_tol=0.05, score_tol=1e-4):
    """results.txt rows `image x1 y1 x2 y2 score class` of two detectors -> (matched, unmatched_a, unmatched_b): a row of A is matched
    by an unused row of B of the same image and class whose four coordinates differ by <= box_tol px and whose score by <= score_tol"""
    def parse(rows):
        d = {}
        for r in rows:
            f = r.split()
            d.setdefault((f[0], f[6]), []).append(np.array(f[1:6], dtype=np.float64))
        return d
    A, B = parse(rows_a), parse(rows_b)
    matched = un_a = 0
    for key, la in A.items():
        lb = list(B.get(key, []))
        for a in la:
            hit = next((i for i, b in enumerate(lb) if np.abs(a[:4] - b[:4]).max() <= box_tol and abs(a[4] - b[4]) <= score_tol), None)
            if hit is None:
                un_a += 1
            else:
                matched += 1
                lb.pop(hit)
    return matched, un_a, sum(len(v) for v in B.values()) - matched


def run(cuda, workdir, iters=400, lr=1e-4, verbose=False, eval_seed=None):
    """eval_seed: evaluate on a HELD-OUT set drawn with that seed (other rectangles, other positions) instead of the training images"""
    from oracle import torch_ref as R
    from scda_amd.evaluate import validate
    images, gts, names = make_dataset()
    tr, hist = train_on_device(cuda, images, gts, iters, lr)
    if eval_seed is not None:
        images, gts, names = make_dataset(eval_seed)
        names = [n.replace("synth_", "held_") for n in names]
    meta = os.path.join(workdir, "val_meta.txt")
    with open(meta, "w") as f:
        f.writelines(meta_lines(names, gts))
    if verbose:
        for h in hist:
            print("iter %4d  rpn_cls %.4f rpn_loc %.4f rcnn_cls %.4f rcnn_loc %.4f rcnn_acc %.1f" % h)
    state = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}          # THE checkpoint both sides evaluate
    # (b) HIP detector
    d_hip = os.path.join(workdir, "hip")
    rc_hip = validate(loader(images, gts, names), tr.model, CFG, d_hip, score=False)
    map_hip, rows_hip = score(d_hip, meta)
    # (a) CPU oracle detector, same weights
    torch.manual_seed(1)
    ref = R.build_models(CFG)[0]
    missing = ref.load_state_dict(state, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    d_ref = os.path.join(workdir, "oracle")
    R.use_cpu_backend()
    try:
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        rc_ref = validate(loader(images, gts, names), ref, CFG, d_ref, score=False)
    finally:
        R.reset_backend()
        torch.set_num_threads(1)
    map_ref, rows_ref = score(d_ref, meta)
    matched, only_hip, only_ref = compare_rows(rows_hip, rows_ref)
    return dict(map_hip=map_hip, map_ref=map_ref, rows_hip=len(rows_hip), rows_ref=len(rows_ref), recall_hip=rc_hip, recall_ref=rc_ref,
                rows_matched=matched, rows_only_hip=only_hip, rows_only_ref=only_ref, hist=hist)


@pytest.mark.gpu
def test_map_of_one_checkpoint_hip_vs_oracle(cuda, tmp_path):
    r = run(cuda, str(tmp_path))
    print("mAP@0.5 of the same checkpoint: HIP detector %.3f, CPU oracle detector %.3f (rows %d / %d, RPN recall %.3f / %.3f)"
          % (r["map_hip"], r["map_ref"], r["rows_hip"], r["rows_ref"], r["recall_hip"], r["recall_ref"]))
    assert r["map_ref"] > 30.0 and r["map_hip"] > 30.0, r          # a detector that detects: the comparison is not 0 == 0
    assert abs(r["map_hip"] - r["map_ref"]) <= 0.3, r              # north_star: within +-0.3 mAP points
    assert r["rows_hip"] == r["rows_ref"], r
    assert abs(r["recall_hip"] - r["recall_ref"]) <= 1.0 / (N_IMG * PER_IMG) + 1e-9, r


@pytest.mark.gpu
def test_map_rows_of_an_unsaturated_checkpoint_on_held_out_images(cuda, tmp_path):
    """The 400-iteration checkpoint above scores 99 on the images it was trained on: a +-0.3 gate at 99 cannot see a box-level
    regression.  Here training stops at 300 iterations and the checkpoint is evaluated on HELD-OUT images (other rectangles, other
    positions, seed 11): mAP@0.5 around 60 -- and besides the mAP the `results.txt` ROWS are compared: every row of one detector must
    have its partner in the other's file (same image, same class, |box| <= 0.05 px, |score| <= 1e-4)."""
    r = run(cuda, str(tmp_path), iters=300, eval_seed=11)
    print("held-out mAP@0.5: HIP %.3f, CPU oracle %.3f; rows %d / %d, matched %d, only HIP %d, only oracle %d"
          % (r["map_hip"], r["map_ref"], r["rows_hip"], r["rows_ref"], r["rows_matched"], r["rows_only_hip"], r["rows_only_ref"]))
    assert 30.0 < r["map_ref"] < 85.0 and 30.0 < r["map_hip"] < 85.0, r        # neither trivial nor saturated
    assert abs(r["map_hip"] - r["map_ref"]) <= 0.3, r
    assert r["rows_hip"] == r["rows_ref"], r
    assert r["rows_matched"] >= 0.99 * r["rows_ref"], r       # a tie broken the other way may swap a pair of low-score rows


@pytest.mark.gpu
def test_map_rows_of_an_unsaturated_checkpoint_at_the_metric_size(cuda, tmp_path):
    """The same comparison at the size the metric is quoted on (BASELINE.json configs[1]: 512 x 1024): trained on the device at that
    size, the checkpoint evaluated on held-out 512 x 1024 images by both detectors -- mAP within 0.3, rows matched >= 99 %."""
    set_size(512, 1024)
    try:
        r = run(cuda, str(tmp_path), iters=FULL_ITERS, eval_seed=11)
    finally:
        set_size(256, 512)
    print("512x1024 held-out mAP@0.5: HIP %.3f, CPU oracle %.3f; rows %d / %d, matched %d, only HIP %d, only oracle %d"
          % (r["map_hip"], r["map_ref"], r["rows_hip"], r["rows_ref"], r["rows_matched"], r["rows_only_hip"], r["rows_only_ref"]))
    assert 30.0 < r["map_ref"] < 90.0 and 30.0 < r["map_hip"] < 90.0, r        # neither trivial nor saturated
    assert abs(r["map_hip"] - r["map_ref"]) <= 0.3, r
    assert r["rows_hip"] == r["rows_ref"], r
    assert r["rows_matched"] >= 0.99 * r["rows_ref"], r


FULL_ITERS = 400      # mAP 54 on the held-out set (300: 30, 200: 20 -- scripts: python tests/test_map_parity_gpu.py 400 1e-4 11 512)


if __name__ == "__main__":
    import tempfile
    it = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-4
    ev = int(sys.argv[3]) if len(sys.argv) > 3 else None
    if len(sys.argv) > 4:
        set_size(int(sys.argv[4]), 2 * int(sys.argv[4]))
    with tempfile.TemporaryDirectory() as d:
        r = run(torch.device("cuda:0"), d, it, lr, verbose=True, eval_seed=ev)
    print({k: v for k, v in r.items() if k != "hist"})
