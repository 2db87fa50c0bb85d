"""bilinear x2 (align_corners) forward / backward on the decoders' two shapes: time per launch and HBM rate (HIP events, 200 launches)
    [SCDA_UPSAMPLE_BWD_ROWWISE=1] python scripts/bench_upsample.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from scda_amd import native as N
dev = torch.device("cuda:0")
for shape in ((4, 128, 64, 64), (4, 64, 128, 128)):
    x = torch.randn(*shape, device=dev); dy = torch.randn(shape[0], shape[1], 2 * shape[2], 2 * shape[3], device=dev)
    for name, fn, arg in (("fwd", N.upsample2x_fwd, x), ("bwd", N.upsample2x_bwd, dy)):
        for _ in range(10): fn(arg)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(200): fn(arg)
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 200 * 1e3
        byts = (x.numel() + dy.numel()) * 4
        print("%s %-18s %6.1f us  %5.2f TB/s%s" % (name, tuple(shape), us, byts / us / 1e6, "  (row-at-a-time backward)" if name == "bwd" and os.environ.get("SCDA_UPSAMPLE_BWD_ROWWISE") else ""))
