import os, sys
sys.path.insert(0, "/root/repo")
import torch
from scda_amd import native as N
dev = torch.device("cuda:0")
def t(fn, it=10):
    fn(); torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / it * 1e3
os.environ["SCDA_PLAN_ALLOW_BM64"] = "1"
for name, B, Cin, H, W, Cout in [("dec_res", 4, 128, 64, 64, 128), ("conv5_x", 1, 512, 32, 64, 512), ("rpn3x3", 1, 512, 32, 64, 512), ("conv2_1", 1, 64, 256, 512, 128)]:
    x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
    y = N.conv2d_fwd(x, w, b, 1, 1, 1); dy = torch.randn_like(y)
    ref = y.clone()
    for op, fn in (("fwd", lambda: N.conv2d_fwd(x, w, b, 1, 1, 1)), ("dgrad", lambda: N.conv2d_dgrad(dy, w, x.shape, 1, 1))):
        os.environ.pop("SCDA_PLAN_FORCE", None)
        base = t(fn)
        res = []
        for c in [(64, 64, 1), (64, 128, 1), (64, 256, 1), (64, 64, 2), (64, 128, 2)]:
            os.environ["SCDA_PLAN_FORCE"] = "%d,%d,%d" % c
            res.append((round(t(fn), 1), c))
        if op == "fwd":
            os.environ["SCDA_PLAN_FORCE"] = "64,64,1"
            err = float((N.conv2d_fwd(x, w, b, 1, 1, 1) - ref).abs().max())
        os.environ.pop("SCDA_PLAN_FORCE", None)
        print(name, op, "model %.1f us" % base, sorted(res)[:3], "maxerr %.2e" % err)
