"""The device data path (scda_image_resize_normalize_hip through scda_amd/device_image.py and data.py's `device=`):
bit-identical to PIL's resize + flip + ToTensor + Normalize, and to the reference's own dataset outputs (tests/golden/data_path.npz)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from test_data_path import _materialise  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cpu_path(a, new_w, new_h, flip, do_norm=True, pil_filter=None):
    from PIL import Image
    from scda_amd import data as D
    img = Image.fromarray(a if a.shape[-1] != 1 else a[:, :, 0])
    img = img.resize((new_w, new_h)) if pil_filter is None else img.resize((new_w, new_h), pil_filter)
    if flip:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)
    t = D.to_tensor(img)
    return D.normalize(t) if do_norm else t


@pytest.mark.parametrize("H,W,nh,nw,C", [(64, 128, 32, 64, 3), (37, 91, 50, 120, 3), (100, 60, 100, 33, 3), (48, 48, 96, 48, 3),
                                         (128, 256, 75, 150, 3), (20, 30, 7, 11, 3), (33, 65, 33, 65, 3), (40, 72, 31, 50, 1),
                                         (1024, 2048, 512, 1024, 3), (600, 1200, 800, 1600, 3)])
@pytest.mark.parametrize("flip", [False, True])
def test_resize_flip_to_tensor_normalize_equal_pil(H, W, nh, nw, C, flip):
    from scda_amd import device_image as DI
    rng = np.random.default_rng(H * 7 + W)
    a = rng.integers(0, 256, (H, W, C), dtype=np.uint8)
    got = DI.resize_to_tensor(a, nw, nh, DEV, flip=flip)
    ref = _cpu_path(a, nw, nh, flip)
    assert got.shape == ref.shape and got.dtype == torch.float32
    assert torch.equal(got.cpu(), ref)


def test_saturated_image_and_no_normalize():
    from scda_amd import device_image as DI
    a = np.zeros((50, 70, 3), np.uint8); a[:, ::3] = 255; a[::4, :, 1] = 255
    for nh, nw in ((33, 120), (80, 31)):
        got = DI.resize_to_tensor(a, nw, nh, DEV, normalize=False)
        assert torch.equal(got.cpu(), _cpu_path(a, nw, nh, False, do_norm=False))
        assert float(got.min()) >= 0.0 and float(got.max()) <= 1.0


def test_other_filters_follow_their_tables():
    from PIL import Image
    from scda_amd import device_image as DI
    rng = np.random.default_rng(9)
    a = rng.integers(0, 256, (90, 140, 3), dtype=np.uint8)
    for name, pf in (("bilinear", Image.BILINEAR), ("lanczos", Image.LANCZOS), ("box", Image.BOX), ("hamming", Image.HAMMING)):
        got = DI.resize_to_tensor(a, 77, 41, DEV, filter=name)
        assert torch.equal(got.cpu(), _cpu_path(a, 77, 41, False, pil_filter=pf)), name


def test_bad_arguments_are_errors():
    from scda_amd import native as N
    from scda_amd import device_image as DI
    for c in (2, 4):      # only L and RGB: PIL pre-multiplies alpha modes before it resizes them
        a = torch.zeros(8, 8, c, dtype=torch.uint8, device=DEV)
        with pytest.raises(N.ScdaNativeError):
            N.image_resize_normalize(a, DI.resize_tables(8, 8, 4, 4, DEV), 4, 4)
    with pytest.raises(N.ScdaNativeError):                        # a host tensor: there is no CPU path
        N.image_resize_normalize(torch.zeros(8, 8, 3, dtype=torch.uint8), DI.resize_tables(8, 8, 4, 4, DEV), 4, 4)


def test_datasets_on_the_device_equal_the_reference_items(tmp_path):
    """same files, same numpy seed as tests/golden/make_golden_data.py ran the REFERENCE's datasets with: identical tensors"""
    from scda_amd import data as D
    z, names = _materialise(tmp_path)
    ds = D.ExampleDataset(str(tmp_path), str(tmp_path / "train_list.txt"), D.ExampleTransform([48, 64], 100, flip=True), device=DEV)
    np.random.seed(5)
    items = [ds[i] for i in (0, 1, 2, 0)]
    for k, it in enumerate(items):
        assert it[0].is_cuda and torch.equal(it[0].cpu(), torch.from_numpy(z["item%d_img" % k])), k
        assert torch.equal(it[1], torch.from_numpy(z["item%d_info" % k])), k
        assert torch.equal(it[2], torch.from_numpy(z["item%d_gt" % k])) and torch.equal(it[3], torch.from_numpy(z["item%d_ig" % k])), k
    img, info, gt, ig, fn = D.collate(items[:3])
    assert img.is_cuda and torch.equal(img.cpu(), torch.from_numpy(z["batch_img"])) and torch.equal(gt, torch.from_numpy(z["batch_gt"]))
    td = D.TargetDataset(str(tmp_path), str(tmp_path / "target_list.txt"), new_w=48, new_h=24, device=DEV)
    for k in range(3):
        assert torch.equal(td[k].cpu(), torch.from_numpy(z["target%d" % k])), k
    cfg = {'shared': {'scales': [48], 'max_size': 100}}
    with pytest.raises(ValueError):
        D.build_data_loaders(str(tmp_path), str(tmp_path / "train_list.txt"), str(tmp_path / "train_list.txt"),
                             str(tmp_path / "target_list.txt"), cfg, workers=2, device=DEV)
    tl, vl, gl = D.build_data_loaders(str(tmp_path), str(tmp_path / "train_list.txt"), str(tmp_path / "train_list.txt"),
                                      str(tmp_path / "target_list.txt"), cfg, batch_size=2, new_w=48, new_h=24, device=DEV)
    np.random.seed(0); torch.manual_seed(0)
    b = next(iter(tl))
    assert b[0].is_cuda and b[0].shape[0] == 2
    assert tuple(next(iter(gl)).shape) == (2, 3, 24, 48)
