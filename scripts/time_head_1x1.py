import os, sys
sys.path.insert(0, "/root/repo")
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
for name, (B, Cin, H, W, Cout) in {"p4_512_2048": (1, 512, 3584, 7, 2048), "p4_2048_512": (1, 2048, 3584, 7, 512), "p4_1024_512": (1, 1024, 3584, 7, 512),
                                    "p3_256_1024": (1, 256, 50, 84, 1024), "p2_128_512": (1, 128, 100, 168, 512)}.items():
    x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, 1, 1, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
    y = native.conv2d_fwd(x, w, b, 1, 0, 1); dy = torch.randn_like(y)
    tf = timeit(lambda: native.conv2d_fwd(x, w, b, 1, 0, 1)); pf = native.last_plan()
    td = timeit(lambda: native.conv2d_dgrad(dy, w, x.shape, 1, 0)); pd = native.last_plan()
    fl = 2.0 * B * H * W * Cin * Cout
    print("%-12s fwd %7.1f us %6.1f TF %s | dgrad %7.1f us %6.1f TF %s" % (name, tf, fl / tf / 1e6, pf, td, fl / td / 1e6, pd))
