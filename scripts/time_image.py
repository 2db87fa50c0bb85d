"""Data path per image: PIL resize + ToTensor + Normalize on the host against scda_image_resize_normalize_hip,
for a Cityscapes frame (1024 x 2048 -> 512 x 1024, the reference's training size) and the ResNet configuration's 800 x 1600."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from scda_amd import data as D, device_image as DI, native as N

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for H, W, nh, nw in ((1024, 2048, 512, 1024), (1024, 2048, 800, 1600)):
    a = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    img = Image.fromarray(a)
    t0 = time.perf_counter()
    for _ in range(5):
        ref = D.normalize(D.to_tensor(img.resize((nw, nh))))
    t_cpu = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(5):
        ref_dev = ref.to(dev)
    torch.cuda.synchronize(); t_up = (time.perf_counter() - t0) / 5
    got = DI.resize_to_tensor(a, nw, nh, dev); torch.cuda.synchronize()
    assert torch.equal(got.cpu(), ref)
    t0 = time.perf_counter()
    for _ in range(20):
        got = DI.resize_to_tensor(a, nw, nh, dev)
    torch.cuda.synchronize(); t_dev = (time.perf_counter() - t0) / 20
    src = N.upload(a, dev); tab = DI.resize_tables(H, W, nh, nw, dev)
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    N.image_resize_normalize(src, tab, nh, nw); torch.cuda.synchronize()
    s.record()
    for _ in range(50):
        N.image_resize_normalize(src, tab, nh, nw)
    e.record(); torch.cuda.synchronize()
    t_k = s.elapsed_time(e) / 50 * 1e-3
    rows = tab[7]
    byts = rows * W * 3 + 2 * rows * nw * 3 + nh * nw * 3 * 4      # source rows read, intermediate written + read, float planes written
    print("%dx%d -> %dx%d | host PIL + ToTensor + Normalize %.1f ms (+ %.2f ms pageable upload of the float tensor) | device path incl. pinned "
          "upload of the bytes %.2f ms | the two launches %.1f us = %.0f GB/s of %.1f MB algorithmic | identical: True"
          % (H, W, nh, nw, t_cpu * 1e3, t_up * 1e3, t_dev * 1e3, t_k * 1e6, byts / t_k / 1e9, byts / 1e6))
