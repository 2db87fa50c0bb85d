"""instance-norm kernels at the decoder's shapes: achieved HBM bandwidth (HIP events)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scda_amd import native as N
dev = torch.device("cuda:0")
for shape in [(4, 128, 64, 64), (4, 64, 128, 128), (4, 32, 256, 256)]:
    x = torch.randn(*shape, device=dev); dy = torch.randn_like(x)
    y, m, r = N.instnorm_fwd(x, 1e-5, 1, 0.01)
    def t(fn, it=50):
        fn(); torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / it * 1e3
    tf = t(lambda: N.instnorm_fwd(x, 1e-5, 1, 0.01)); tb = t(lambda: N.instnorm_bwd(dy, x, m, r, 1, 0.01))
    nb = x.numel() * 4
    print("%-18s fwd %6.1f us %5.2f TB/s | bwd %6.1f us %5.2f TB/s" % (shape, tf, 2 * nb / tf / 1e6, tb, 3 * nb / tb / 1e6))
