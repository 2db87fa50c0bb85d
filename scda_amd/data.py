"""Data path of the reference driver (datasets/example_dataset.py, example_loader.py, target_dataset.py), without the
torchvision dependency: meta-file parsing, the resize / floor-ceil box scaling / flip transform with the reference's numpy
RNG calls, PIL decoding, [-1,1] normalisation, zero-padding collate.

Meta file (Cityscapes-style detection list, one record per image, example_dataset.py:39-69):
    # <index>
    <path relative to the data dir>
    <unused>
    <height>
    <width>
    <unused>
    <number of ignore regions n_ig>
    n_ig lines  "x1 y1 x2 y2"
    <number of ground-truth boxes n_gt>
    n_gt lines  "label x1 y1 x2 y2"
The target-domain list is one relative image path per line (target_dataset.py:31-33).
"""
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import DataLoader, Dataset


def parse_meta(list_file):
    """-> list of [name, height, width, gt [G,4] float64, labels [G] int, ignores [I,4] float64]"""
    with open(list_file) as f:
        lines = f.readlines()
    metas, i = [], 0
    while i < len(lines):
        name = lines[i + 1].rstrip()
        height, width = float(lines[i + 3]), float(lines[i + 4])
        n_ig = int(lines[i + 6])
        i += 7
        ig = [[float(v) for v in lines[i + j].split()[:4]] for j in range(n_ig)] or [[0, 0, 0, 0]]
        i += n_ig
        n_gt = int(lines[i])
        i += 1
        rows = [lines[i + j].split() for j in range(n_gt)]
        gt = [[float(r[1]), float(r[2]), float(r[3]), float(r[4])] for r in rows]
        labels = [int(r[0]) for r in rows]
        i += n_gt
        metas.append([name, height, width, np.array(gt), np.array(labels), np.array(ig)])
    return metas


def to_tensor(img):
    """PIL RGB image -> float32 [3,H,W] in [0,1]  (torchvision.transforms.ToTensor)"""
    a = np.asarray(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div(255)


def normalize(t, mean=0.5, std=0.5):
    """(x - 0.5) / 0.5 per channel (faster_rcnn_train_val.py:195)"""
    return (t - mean) / std


class ExampleTransform(object):
    """random short-side scale in [min(sizes), max(sizes)] capped by max_size, boxes floor/ceil'd, optional horizontal flip
    (example_dataset.py:107-145); consumes np.random.randint, then np.random.random when flip is on, like the reference"""

    def __init__(self, sizes, max_size, flip=False):
        sizes = sizes if isinstance(sizes, list) else [sizes]
        self.scale_min, self.scale_max, self.max_size, self.flip = min(sizes), max(sizes), max_size, flip

    @staticmethod
    def _scale_boxes(b, scale):
        b = np.array(b)
        if b.shape[0] > 0:
            b[:, 0:2] = np.floor(b[:, 0:2] * scale)
            b[:, 2:4] = np.ceil(b[:, 2:4] * scale)
        return b

    def plan(self, w, h):
        """the transform's random decisions for a w x h image, drawn in the reference's order (randint, then random when flip is
        on): (new_w, new_h, scale, flipped)"""
        size = np.random.randint(self.scale_min, self.scale_max + 1)
        scale = min(size / min(w, h), self.max_size / max(w, h))
        new_w, new_h = int(w * scale), int(h * scale)
        flipped = bool(self.flip and np.random.random() < 0.5)
        return new_w, new_h, scale, flipped

    def boxes(self, bbox, ignores, scale, new_w, flipped):
        bbox = self._scale_boxes(bbox, scale)
        ignores = self._scale_boxes(ignores, scale)
        if flipped:
            bbox[:, 0], bbox[:, 2] = new_w - bbox[:, 2], new_w - bbox[:, 0]
            if ignores.shape[0] > 0:
                ignores[:, 0], ignores[:, 2] = new_w - ignores[:, 2], new_w - ignores[:, 0]
        return bbox, ignores

    def __call__(self, img, bbox, ignores):
        from PIL import Image
        w, h = img.size
        new_w, new_h, scale, flipped = self.plan(w, h)
        img = img.resize((new_w, new_h))
        if flipped:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        bbox, ignores = self.boxes(bbox, ignores, scale, new_w, flipped)
        return img, bbox, scale, ignores


def _device_image(img, new_w, new_h, device, normalize_fn, flipped=False):
    """resize + flip + ToTensor + Normalize of one decoded image on the device (device_image.py): what the lines
    `to_tensor(img.resize(..)[.transpose(..)])`, `normalize_fn(t)` produce, bit for bit, from the image's bytes"""
    from . import device_image
    if normalize_fn is not None and normalize_fn is not normalize:
        raise ValueError("the device data path applies this module's normalize ((x - 0.5) / 0.5) or none; got another normalize_fn")
    return device_image.resize_to_tensor(img, new_w, new_h, device, normalize=normalize_fn is not None, flip=flipped)


class ExampleDataset(Dataset):
    """item = [image [1,3,h,w] in [-1,1], tensor([h, w, scale]), boxes [G,5] (x1,y1,x2,y2,label), ignores [I,4], filename]
    device given: the image is resized / flipped / converted / normalised ON that device from its decoded bytes (identical values; the
    item's image is a device tensor) -- in the loading process itself, so use it with num_workers = 0"""

    def __init__(self, root_dir, list_file, transform_fn, normalize_fn=normalize, device=None):
        self.root_dir, self.transform_fn, self.normalize_fn, self.device = root_dir, transform_fn, normalize_fn, device
        self.metas = parse_meta(list_file)
        self.num = len(self.metas)
        self.aspect_ratios = [float(m[1]) / m[2] for m in self.metas]

    def __len__(self):
        return self.num

    def __getitem__(self, idx):
        from PIL import Image
        filename = os.path.join(self.root_dir, self.metas[idx][0])
        h, w, bbox, labels, ignores = self.metas[idx][1:]
        bbox, ignores, labels = bbox.astype(np.float32), ignores.astype(np.float32), labels.astype(np.float32)
        img = Image.open(filename)
        if img.mode == 'L':
            img = img.convert('RGB')
        assert img.size[0] == w and img.size[1] == h, "image size differs from the meta file"
        if self.device is not None:
            new_w, new_h, scale, flipped = self.transform_fn.plan(w, h)
            bbox, ignores = self.transform_fn.boxes(bbox, ignores, scale, new_w, flipped)
            t = _device_image(img, new_w, new_h, self.device, self.normalize_fn, flipped)
        else:
            img, bbox, scale, ignores = self.transform_fn(img, bbox, ignores)
            new_w, new_h = img.size
            t = to_tensor(img)
            if self.normalize_fn is not None:
                t = self.normalize_fn(t)
        bbox = np.hstack([bbox.reshape(-1, 4), labels[:, np.newaxis]])
        return [t.unsqueeze(0), torch.Tensor([new_h, new_w, scale]), torch.from_numpy(bbox), torch.from_numpy(ignores), filename]


class TargetDataset(Dataset):
    """unlabelled target-domain images, resized to exactly new_w x new_h (target_dataset.py:23-71)"""

    def __init__(self, root_dir, list_file, normalize_fn=normalize, new_w=1024, new_h=512, device=None):
        self.root_dir, self.normalize_fn, self.new_w, self.new_h, self.device = root_dir, normalize_fn, new_w, new_h, device
        with open(list_file) as f:
            self.metas = [x.strip() for x in f.readlines()]
        self.num = len(self.metas)

    def __len__(self):
        return self.num

    def __getitem__(self, idx):
        from PIL import Image
        img = Image.open(os.path.join(self.root_dir, self.metas[idx]))
        if img.mode == 'L':
            img = img.convert('RGB')
        if self.device is not None:
            return _device_image(img, self.new_w, self.new_h, self.device, self.normalize_fn)
        t = to_tensor(img.resize((self.new_w, self.new_h)))
        return self.normalize_fn(t) if self.normalize_fn is not None else t


def collate(batch):
    """images zero-padded at the right/bottom to the batch maximum, boxes / ignores zero-padded to the longest list
    (example_loader.py:13-56): (images [B,3,H,W], sizes [B,3], gts float64 [B,G,5], ignores float64 [B,I,4], filenames)"""
    images, sizes, gts, igs, names = list(zip(*batch))
    H, W = max(t.shape[-2] for t in images), max(t.shape[-1] for t in images)
    G, I = max(g.shape[0] for g in gts), max(g.shape[0] for g in igs)
    pad_img = [F.pad(t, (0, W - t.shape[-1], 0, H - t.shape[-2]), 'constant', 0) for t in images]

    def pad_rows(rows, n):
        out = np.zeros([len(rows), n, rows[0].shape[-1]])
        for b, r in enumerate(rows):
            out[b, :r.shape[0], :] = r.numpy()
        return torch.from_numpy(out)

    return torch.cat(pad_img, dim=0), torch.stack(sizes, dim=0), pad_rows(gts, G), pad_rows(igs, I), names


class ExampleDataLoader(DataLoader):
    def __init__(self, dataset, batch_size=1, shuffle=False, sampler=None, batch_sampler=None, num_workers=0,
                 pin_memory=False, drop_last=False):
        super().__init__(dataset, batch_size=batch_size, shuffle=shuffle, sampler=sampler, batch_sampler=batch_sampler,
                         num_workers=num_workers, collate_fn=collate, pin_memory=pin_memory, drop_last=drop_last)


def build_data_loaders(datadir, train_meta_file, val_meta_file, target_meta_file, cfg, batch_size=1, workers=0,
                       distributed=False, new_w=1024, new_h=512, device=None):
    """(train_loader, val_loader, target_loader) as the reference's build_data_loader (faster_rcnn_train_val.py:191-248).
    device given: images are resized and normalised on that device (same values; batches arrive as device tensors)"""
    from torch.utils.data.distributed import DistributedSampler
    if device is not None and workers != 0:
        raise ValueError("the device data path runs in the loading process: workers must be 0")
    scales, max_size = cfg['shared']['scales'], cfg['shared']['max_size']
    train = ExampleDataset(datadir, train_meta_file, ExampleTransform(scales, max_size, flip=True), device=device)
    val = ExampleDataset(datadir, val_meta_file, ExampleTransform(max(scales), max_size, flip=False), device=device)
    target = TargetDataset(datadir, target_meta_file, new_w=new_w, new_h=new_h, device=device)
    ts, vs, gs = (DistributedSampler(d) for d in (train, val, target)) if distributed else (None, None, None)
    return (ExampleDataLoader(train, batch_size=batch_size, shuffle=ts is None, num_workers=workers, sampler=ts),
            ExampleDataLoader(val, batch_size=1, shuffle=False, num_workers=workers, sampler=vs),
            DataLoader(target, batch_size=batch_size, shuffle=gs is None, num_workers=workers, sampler=gs))
