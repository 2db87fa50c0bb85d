"""isolated timing of the stride-2 data-gradient launches of the discriminators (us, TFLOP/s of necessary work)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scda_amd import native as N
dev = torch.device("cuda:0")
SH = [("dis l2  32<-64", 4, 32, 128, 128, 64), ("dis l3  64<-128", 4, 64, 64, 64, 128), ("patch l1 128<-256", 4, 128, 64, 64, 256),
      ("patch l2 256<-512", 4, 256, 32, 32, 512), ("patch l3 512<-512", 4, 512, 16, 16, 512)]
def t(fn, it=20):
    fn(); torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / it * 1e3
for name, B, Cin, H, W, Cout in SH:
    x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
    y = N.conv2d_fwd(x, w, None, 2, 1, 0); dy = torch.randn_like(y)
    for label, fn in (("fwd", lambda: N.conv2d_fwd(x, w, None, 2, 1, 0)), ("dgrad", lambda: N.conv2d_dgrad(dy, w, x.shape, 2, 1)),
                      ("wgrad", lambda: N.conv2d_wgrad(dy, x, w.shape, 2, 1))):
        us = t(fn); fl = 2.0 * Cout * Cin * 9 * y.shape[0] * y.shape[2] * y.shape[3]
        print("%-18s %-6s %7.1f us  %6.1f TF/s  plan %s" % (name, label, us, fl / us / 1e6, N.last_plan()[:3]))
