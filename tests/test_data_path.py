"""Data path (scda_amd/data.py) against the reference's own datasets/*.py outputs (tests/golden/data_path.npz, made by
tests/golden/make_golden_data.py): same files, same numpy seed -> identical tensors."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from conftest import GOLDEN  # noqa: E402


def _materialise(tmp_path):
    z = np.load(os.path.join(GOLDEN, "data_path.npz"))
    names = [str(n) for n in z["names"]]
    for n in names:
        p = tmp_path / n
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(z["png_" + n.replace("/", "__")].tobytes())
    (tmp_path / "train_list.txt").write_text(str(z["meta"]))
    (tmp_path / "target_list.txt").write_text("".join(n + "\n" for n in names))
    return z, names


def test_dataset_transform_collate_match_reference(tmp_path):
    from scda_amd import data as D
    z, names = _materialise(tmp_path)
    metas = D.parse_meta(str(tmp_path / "train_list.txt"))
    assert [m[0] for m in metas] == names and [len(m[4]) for m in metas] == [3, 1, 4]
    assert metas[0][5].tolist() == [[0, 0, 0, 0]]                 # no ignore region -> the reference's single zero row
    ds = D.ExampleDataset(str(tmp_path), str(tmp_path / "train_list.txt"), D.ExampleTransform([48, 64], 100, flip=True))
    assert len(ds) == 3 and abs(ds.aspect_ratios[0] - 60 / 100) < 1e-12
    np.random.seed(5)
    items = [ds[i] for i in (0, 1, 2, 0)]
    for k, it in enumerate(items):
        assert torch.equal(it[0], torch.from_numpy(z["item%d_img" % k])), k
        assert torch.equal(it[1], torch.from_numpy(z["item%d_info" % k])), k
        assert it[2].dtype == torch.float32 and torch.equal(it[2], torch.from_numpy(z["item%d_gt" % k])), k
        assert torch.equal(it[3], torch.from_numpy(z["item%d_ig" % k])), k
        assert it[4] == os.path.join(str(tmp_path), names[(0, 1, 2, 0)[k]])
    img, info, gt, ig, fn = D.collate(items[:3])
    assert torch.equal(img, torch.from_numpy(z["batch_img"])) and torch.equal(info, torch.from_numpy(z["batch_info"]))
    assert gt.dtype == torch.float64 and torch.equal(gt, torch.from_numpy(z["batch_gt"]))
    assert torch.equal(ig, torch.from_numpy(z["batch_ig"]))
    assert list(fn) == [it[4] for it in items[:3]]
    td = D.TargetDataset(str(tmp_path), str(tmp_path / "target_list.txt"), new_w=48, new_h=24)
    for k in range(3):
        assert torch.equal(td[k], torch.from_numpy(z["target%d" % k])), k


def test_loaders_iterate(tmp_path):
    from scda_amd import data as D
    _materialise(tmp_path)
    cfg = {'shared': {'scales': [48], 'max_size': 100}}
    tl, vl, gl = D.build_data_loaders(str(tmp_path), str(tmp_path / "train_list.txt"), str(tmp_path / "train_list.txt"),
                                      str(tmp_path / "target_list.txt"), cfg, batch_size=2, new_w=48, new_h=24)
    np.random.seed(0); torch.manual_seed(0)
    b = next(iter(tl))
    assert b[0].shape[0] == 2 and b[0].dtype == torch.float32 and b[2].shape[0] == 2 and b[2].shape[2] == 5
    assert float(b[0].min()) >= -1.0 and float(b[0].max()) <= 1.0
    v = list(vl)
    assert len(v) == 3 and all(x[0].shape[0] == 1 for x in v)
    g = next(iter(gl))
    assert tuple(g.shape) == (2, 3, 24, 48)
