"""1x1 stride-1 convolutions of a batch-1 detector ARE dense GEMMs (out[Cout][HW] = W[Cout][Cin] X[Cin][HW]): the convolution kernels'
plan against scda_gemm_hip's on the ResNet-50 C4 shapes, forward and data gradient, with the results compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scda_amd import native
dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
shapes = {"p4_512_2048": (512, 3584, 7, 2048), "p4_2048_512": (2048, 3584, 7, 512), "p4_1024_512": (1024, 3584, 7, 512), "p4_1024_2048": (1024, 3584, 7, 2048),
          "p3_256_1024": (256, 50, 84, 1024), "p3_1024_256": (1024, 50, 84, 256), "p3_512_256": (512, 50, 84, 256), "p2_128_512": (128, 100, 168, 512), "p2_512_128": (512, 100, 168, 128),
          "p2_256_128": (256, 100, 168, 128), "rpn_512_30": (512, 50, 84, 30)}
for name, (Cin, H, W, Cout) in shapes.items():
    x = torch.randn(1, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, 1, 1, device=dev) * 0.05
    HW = H * W
    y = native.conv2d_fwd(x, w, None, 1, 0, 0); dy = torch.randn_like(y)
    tf = timeit(lambda: native.conv2d_fwd(x, w, None, 1, 0, 0)); pf = native.last_plan()
    td = timeit(lambda: native.conv2d_dgrad(dy, w, x.shape, 1, 0)); pd = native.last_plan()
    dx = native.conv2d_dgrad(dy, w, x.shape, 1, 0)
    w2 = w.view(Cout, Cin)
    gf = lambda: native.gemm(w2, x, Cout, HW, Cin, Cin, HW, False, True)
    gd = lambda: native.gemm(w2, dy, Cin, HW, Cout, Cin, HW, True, True)
    yg = gf(); pgf = native.last_plan(); dxg = gd(); pgd = native.last_plan()
    ef = (yg.view_as(y) - y).abs().max().item() / y.abs().max().item(); ed = (dxg.view_as(dx) - dx).abs().max().item() / dx.abs().max().item()
    tgf, tgd = timeit(gf), timeit(gd)
    fl = 2.0 * HW * Cin * Cout
    print("%-13s fwd conv %7.1f us %6.1f TF %-18s gemm %7.1f us %6.1f TF %-18s err %.1e | dgrad conv %7.1f us %6.1f TF %-18s gemm %7.1f us %6.1f TF %-18s err %.1e" %
          (name, tf, fl / tf / 1e6, pf, tgf, fl / tgf / 1e6, pgf, ef, td, fl / td / 1e6, pd, tgd, fl / tgd / 1e6, pgd, ed))
