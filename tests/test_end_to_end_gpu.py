"""files on disk -> data loaders -> two training iterations -> checkpoint -> validation pass, all product code on the MI355X"""
import copy
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from test_host_functions import CFG  # noqa: E402

pytestmark = pytest.mark.gpu


def test_files_to_checkpoint_to_validation(cuda, tmp_path):
    from PIL import Image
    from scda_amd import checkpoint, data, evaluate
    from scda_amd.train_step import ScdaTrainer
    cfg = copy.deepcopy(CFG)
    cfg['shared'].update(gan_model_flag=2, scales=[256], max_size=512)
    r = np.random.RandomState(3)
    W, H = 512, 256
    meta, names = [], []
    for i in range(2):
        name = "city/img_%03d_leftImg8bit.png" % i
        (tmp_path / "city").mkdir(exist_ok=True)
        Image.fromarray(r.randint(0, 256, (H, W, 3)).astype(np.uint8), 'RGB').save(tmp_path / name)
        names.append(name)
        meta += ["# %d\n" % i, name + "\n", "3\n", "%d\n" % H, "%d\n" % W, "0\n", "0\n", "3\n"]
        for _ in range(3):
            x1, y1 = r.randint(0, W - 120), r.randint(0, H - 90)
            meta.append("%d %d %d %d %d\n" % (r.randint(1, 9), x1, y1, x1 + r.randint(30, 110), y1 + r.randint(30, 80)))
    (tmp_path / "list.txt").write_text("".join(meta))
    (tmp_path / "target.txt").write_text("".join(n + "\n" for n in names))
    np.random.seed(0); torch.manual_seed(0)
    train, val, target = data.build_data_loaders(str(tmp_path), str(tmp_path / "list.txt"), str(tmp_path / "list.txt"),
                                                 str(tmp_path / "target.txt"), cfg, new_w=W, new_h=H)
    tr = ScdaTrainer(cfg, cuda, lr=1e-4, new_w=W, new_h=H)
    seen = 0
    for (img, info, gts, _, _), tgt in zip(train, target):
        assert tuple(img.shape) == (1, 3, H, W) and tuple(tgt.shape) == (1, 3, H, W)
        out = tr.step(img.to(cuda), gts, info, tgt.to(cuda))
        assert all(np.isfinite(float(v)) for v in out.values() if torch.is_tensor(v) and v.numel() == 1)
        seen += 1
    assert seen == 2
    checkpoint.save_checkpoint(tr, str(tmp_path / "ck.pth"), epoch=1)
    recall = evaluate.validate(val, tr.model, cfg, str(tmp_path / "res"), score=False)
    assert 0.0 <= recall <= 1.0
    rows = (tmp_path / "res" / "results.txt.rank0").read_text().splitlines()
    assert rows and all(len(l.split()) == 7 and l.split()[0].endswith("leftImg8bit") for l in rows)
    assert tr.model.training        # validate() restores the mode it found
