"""Golden vectors for the data path: the reference's own ExampleDataset / ExampleTransform /
ExampleDataLoader._collate_fn / TargetDataset (datasets/*.py, imported unmodified) on three small synthetic PNG files.
torchvision is not installed here; the harness supplies the two transforms the reference calls (ToTensor, Normalize) with
their documented semantics.  Called from make_golden.py --only data.  Output: tests/golden/data_path.npz (inputs = the PNG
bytes and the meta text; outputs = what the reference returned)."""
import importlib
import io
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch


class ToTensor:
    def __call__(self, pic):
        a = np.asarray(pic, dtype=np.uint8)
        if a.ndim == 2:
            a = a[:, :, None]
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div(255)


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

    def __call__(self, t):
        return (t - self.mean) / self.std


def generate(ns, outdir):
    from PIL import Image
    tvt = sys.modules['torchvision.transforms']
    tvt.ToTensor, tvt.Normalize = ToTensor, Normalize
    ED = importlib.import_module("datasets.example_dataset")
    EL = importlib.import_module("datasets.example_loader")
    TD = importlib.import_module("datasets.target_dataset")
    r = np.random.RandomState(77)
    work = tempfile.mkdtemp(prefix="scda_data_")
    cwd = os.getcwd()
    try:
        os.chdir(work)      # the reference drops a pickle of the parsed meta file into the working directory
        specs = [("leftImg8bit/train/a/a_000.png", 100, 60, 'RGB'), ("leftImg8bit/train/b/b_001.png", 90, 68, 'RGB'),
                 ("leftImg8bit/train/b/b_002.png", 80, 50, 'L')]
        pngs, meta = {}, []
        for i, (name, w, h, mode) in enumerate(specs):
            os.makedirs(os.path.dirname(os.path.join(work, name)), exist_ok=True)
            arr = r.randint(0, 256, (h, w) if mode == 'L' else (h, w, 3)).astype(np.uint8)
            Image.fromarray(arr, mode).save(os.path.join(work, name))
            with open(os.path.join(work, name), 'rb') as f:
                pngs[name] = np.frombuffer(f.read(), dtype=np.uint8)
            n_ig, n_gt = (0, 2, 1)[i], (3, 1, 4)[i]
            meta += ["# %d\n" % i, name + "\n", "3\n", "%d\n" % h, "%d\n" % w, "0\n", "%d\n" % n_ig]
            for _ in range(n_ig):
                x1, y1 = r.randint(0, w // 2), r.randint(0, h // 2)
                meta.append("%d %d %d %d\n" % (x1, y1, x1 + r.randint(5, w // 2), y1 + r.randint(5, h // 2)))
            meta.append("%d\n" % n_gt)
            for _ in range(n_gt):
                x1, y1 = r.randint(0, w // 2), r.randint(0, h // 2)
                meta.append("%d %d %d %d %d\n" % (r.randint(1, 9), x1, y1, x1 + r.randint(5, w // 2), y1 + r.randint(5, h // 2)))
        with open("train_list.txt", "w") as f:
            f.writelines(meta)
        with open("target_list.txt", "w") as f:
            f.writelines(n + "\n" for n, *_ in specs)
        norm = Normalize([0.5, 0.5, 0.5], [0.5, 0.5, 0.5])
        ds = ED.ExampleDataset(work, "train_list.txt", ED.ExampleTransform([48, 64], 100, flip=True), normalize_fn=norm)
        np.random.seed(5)
        items = [ds[i] for i in (0, 1, 2, 0)]          # the 4th draw exercises another scale / flip decision
        batch = EL.ExampleDataLoader._collate_fn(None, items[:3])
        td = TD.TargetDataset(work, "target_list.txt", normalize_fn=norm, new_w=48, new_h=24)
        tgt = [td[i] for i in range(3)]
        out = {"meta": np.array("".join(meta)), "names": np.array([n for n, *_ in specs]), "n_items": np.int64(len(items))}
        for n, b in pngs.items():
            out["png_" + n.replace("/", "__")] = b
        for k, it in enumerate(items):
            out["item%d_img" % k], out["item%d_info" % k] = it[0].numpy(), it[1].numpy()
            out["item%d_gt" % k], out["item%d_ig" % k] = it[2].numpy(), it[3].numpy()
        out["batch_img"], out["batch_info"] = batch[0].numpy(), batch[1].numpy()
        out["batch_gt"], out["batch_ig"] = batch[2].numpy(), batch[3].numpy()
        for k, t in enumerate(tgt):
            out["target%d" % k] = t.numpy()
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)
    np.savez_compressed(os.path.join(outdir, "data_path.npz"), **out)
    print("data_path.npz written:", [tuple(out["item%d_img" % k].shape) for k in range(4)], out["item0_info"], out["item3_info"])
