"""Golden vectors for the evaluation path: the reference's own validate_single()
(tools/faster_rcnn_train_val.py:886-981, imported unmodified) on two seeded synthetic images with the seeded
detector, CPU; plus the reference's utils/cal_mAP.py on (a) that run's results file and (b) a synthetic results
list with real true/false positives, duplicate detections, ties and an image without ground truth for a class.
Called from make_golden.py --only eval.  Output: tests/golden/eval_<H>x<W>.npz  (data only)."""
import importlib
import io
import os
import shutil
import tempfile
from contextlib import redirect_stdout

import numpy as np
import torch

import seeded_init as si

SEEDS = dict(det=11, images=(41, 42), gts=(43, 44), synth=45)
NAMES = ("frankfurt_000000_000294_leftImg8bit", "munster_000001_000019_leftImg8bit")


def meta_lines(names, gts_per_image, H, W):
    """the val meta list format parse_gts() reads (utils/cal_mAP.py:16-47): '# i', path, ., H, W, ., ., n, n box rows"""
    out = []
    for i, (n, g) in enumerate(zip(names, gts_per_image)):
        out += ["# %d\n" % i, "val/city/%s.png\n" % n, "3\n", "%d\n" % H, "%d\n" % W, "0\n", "0\n", "%d\n" % len(g)]
        out += ["%d %d %d %d %d\n" % (b[4], b[0], b[1], b[2], b[3]) for b in g]
    return out


def synth_results(seed, names, gts_per_image, num_classes):
    """detections with known structure: jittered copies of most gts (some twice -> duplicates), random false
    positives, a block of exactly tied scores, every class present at least once"""
    r = np.random.RandomState(seed)
    rows = []
    for n, g in zip(names, gts_per_image):
        for b in g:
            k = r.choice([0, 1, 1, 2])
            for _ in range(k):
                j = r.uniform(-6, 6, 4)
                rows.append((n, b[0] + j[0], b[1] + j[1], b[2] + j[2], b[3] + j[3], r.uniform(0.2, 1.0), int(b[4])))
        for _ in range(25):
            x1, y1 = r.uniform(0, 400), r.uniform(0, 200)
            rows.append((n, x1, y1, x1 + r.uniform(10, 100), y1 + r.uniform(10, 60), r.uniform(0.0, 0.9), int(r.randint(1, num_classes))))
    for c in range(1, num_classes):
        rows.append((names[0], 5.0, 5.0, 30.0, 30.0, 0.5, c))
        rows.append((names[1], 7.0, 5.0, 33.0, 31.0, 0.5, c))
    order = r.permutation(len(rows))
    return ["%s %s %s %s %s %s %d\n" % (rows[i][0], np.float32(rows[i][1]), np.float32(rows[i][2]), np.float32(rows[i][3]),
                                         np.float32(rows[i][4]), np.float32(rows[i][5]), rows[i][6]) for i in order]


def generate(ns, outdir, H=256, W=512, G=8):
    T = ns.T
    C = importlib.import_module("utils.cal_mAP")
    a = T.args
    cfg = T.load_config(a.config)
    nc = int(cfg['shared']['num_classes'])
    torch.manual_seed(1)
    model = ns.vgg.vgg16(pretrained=False, cfg=cfg['shared'])
    si.seeded_reinit(model, SEEDS['det'], 'det')
    imgs = [si.synth_images(s, H, W)[0] for s in SEEDS['images']]
    gts = [si.synth_gts(G, s, H, W) for s in SEEDS['gts']]
    info = torch.tensor([[H, W, 1.0]])
    loader = [(img, info.clone(), g.clone(), ["leftImg8bit/val/city/%s.png" % n]) for img, g, n in zip(imgs, gts, NAMES)]
    work = tempfile.mkdtemp(prefix="scda_eval_")
    try:
        meta = os.path.join(work, "val_meta.txt")
        gl = meta_lines(NAMES, [g[0].numpy() for g in gts], H, W)
        with open(meta, "w") as f:
            f.writelines(gl)
        a.results_dir, a.val_meta_file, a.dataset = os.path.join(work, "res"), meta, 'cityscapes'
        buf = io.StringIO()
        # A randomly initialised detector puts its 100 best boxes into one or two classes, and the reference's cal_mAP
        # raises on a class without detections (np.max of an empty array, utils/cal_mAP.py:130) -- so the mAP call at the
        # end of validate_single is parked for this run and cal_mAP is pinned on the synthetic list below instead.
        calls = []
        T.Cal_MAP = lambda *args: calls.append(args)
        with redirect_stdout(buf), torch.no_grad():
            recall = T.validate_single(loader, model, cfg)
        T.Cal_MAP = C.Cal_MAP
        assert len(calls) == 1
        with open(os.path.join(a.results_dir, "results.txt.rank0")) as f:
            res_lines = f.readlines()
        try:
            C.cal_mAP(C.parse_gts(gl, nc), C.parse_res(res_lines), nc, 0.5)
            empty_class_raises = False
        except ValueError:
            empty_class_raises = True
        sl = synth_results(SEEDS['synth'], NAMES, [g[0].numpy() for g in gts], nc)
        ap_s, mr_s = C.cal_mAP(C.parse_gts(gl, nc), C.parse_res(sl), nc, 0.5)
        map_s = C.Cal_MAP1(sl, gl, nc)
        # (c) a larger list in which every class has ground truth, so that the mean is finite
        names3 = NAMES + ("lindau_000002_000019_leftImg8bit",)
        r = np.random.RandomState(SEEDS['synth'] + 1)
        gts3 = []
        for i in range(3):
            g = si.synth_gts(16, SEEDS['synth'] + 2 + i, 512, 1024)[0].numpy()
            g[:, 4] = (np.arange(16) + r.randint(0, 8)) % (nc - 1) + 1
            gts3.append(g)
        gl3 = meta_lines(names3, gts3, 512, 1024)
        sl3 = synth_results(SEEDS['synth'] + 7, names3, gts3, nc)
        ap_3, mr_3 = C.cal_mAP(C.parse_gts(gl3, nc), C.parse_res(sl3), nc, 0.5)
        map_3 = C.Cal_MAP1(sl3, gl3, nc)
    finally:
        shutil.rmtree(work, ignore_errors=True)
    out = dict(H=np.int64(H), W=np.int64(W), G=np.int64(G), recall=np.float64(recall), results=np.array("".join(res_lines)),
               meta=np.array("".join(gl)), empty_class_raises=np.bool_(empty_class_raises),
               synth_results=np.array("".join(sl)), ap_synth=ap_s, max_recall_synth=mr_s, mAP_synth=np.float64(map_s),
               meta3=np.array("".join(gl3)), synth_results3=np.array("".join(sl3)), ap_synth3=ap_3, max_recall_synth3=mr_3,
               mAP_synth3=np.float64(map_3))
    np.savez_compressed(os.path.join(outdir, f"eval_{H}x{W}.npz"), **out)
    print(f"eval_{H}x{W}.npz written: recall {recall:.4f}, {len(res_lines)} result rows, "
          f"synthetic mAP {map_s:.4f} / {map_3:.4f}")
