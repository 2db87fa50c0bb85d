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
    dy2 = torch.randn(shape[0], shape[1], 2 * shape[2], 2 * shape[3], device=dev)
    _, mean, rstd = N.instnorm_fwd(x, 1e-5, act, 0.01)
    ds = N.upsample2x_bwd(dy2)
    t_ub = timeit(lambda: N.upsample2x_bwd(dy2)); t_nb = timeit(lambda: N.instnorm_bwd(ds, x, mean, rstd, act, 0.01))
    t_fb = timeit(lambda: N.instnorm_up2_bwd(dy2, x, mean, rstd, act, 0.01)); t_fbd = timeit(lambda: N.instnorm_drop_up2_bwd(dy2, x, mean, rstd, 0.5, 1234567))
    print("%s backward: gather %.1f us + norm %.1f us | fused %.1f (tail form, + the residual's gradient: %.1f) us" % (shape, t_ub, t_nb, t_fb, t_fbd), flush=True)
    mb = 4.0 * x.numel() * 5 / 1e6
    print("%s: norm %.1f (+tail %.1f) us, upsample %.1f us | fused %.1f (+tail %.1f) us = %.2f TB/s of x + 4x" %
          (shape, t_n, t_nd, t_u, t_f, t_fd, mb / t_f / 1e6 * 1e6 / 1e6), flush=True)
