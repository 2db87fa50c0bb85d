"""instance norm + bilinear x2 as one launch against the two launches, on the decoders' two shapes (HIP events, us per call):
   python scripts/micro/time_norm_up.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scda_amd import native as N

dev = torch.device("cuda:0")


def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters)
    return best * 1e3


for shape, act in (((4, 128, 64, 64), 0), ((4, 64, 128, 128), 2)):
    x = torch.randn(*shape, device=dev); r = torch.randn(*shape, device=dev)
    t_n = timeit(lambda: N.instnorm_fwd(x, 1e-5, act, 0.01))
    t_nd = timeit(lambda: N.instnorm_drop_add_fwd(x, r, 1e-5, 0.5, 1234567))
    y = N.instnorm_fwd(x, 1e-5, act, 0.01)[0]
    t_u = timeit(lambda: N.upsample2x_fwd(y))
    t_f = timeit(lambda: N.instnorm_up2_fwd(x, 1e-5, act, 0.01))
    t_fd = timeit(lambda: N.instnorm_drop_add_up2_fwd(x, r, 1e-5, 0.5, 1234567))
    mb = 4.0 * x.numel() * 5 / 1e6
    print("%s: norm %.1f (+tail %.1f) us, upsample %.1f us | fused %.1f (+tail %.1f) us = %.2f TB/s of x + 4x" %
          (shape, t_n, t_nd, t_u, t_f, t_fd, mb / t_f / 1e6 * 1e6 / 1e6), flush=True)
