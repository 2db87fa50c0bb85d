"""brute-force the launch plan (tile rows, tile width, split-K) of the workload's conv / FC shapes and compare with what the
occupancy model in conv_gemm.hip picks: how much does the model leave on the table?   (SCDA_PLAN_FORCE=bm,bn,splits)"""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scda_amd import native as N
dev = torch.device("cuda:0")
LAYERS = [("conv1_2", 1, 64, 512, 1024, 64), ("conv2_1", 1, 64, 256, 512, 128), ("conv2_2", 1, 128, 256, 512, 128),
          ("conv3_1", 1, 128, 128, 256, 256), ("conv3_2", 1, 256, 128, 256, 256), ("conv4_1", 1, 256, 64, 128, 512),
          ("conv4_2", 1, 512, 64, 128, 512), ("conv5_x", 1, 512, 32, 64, 512), ("dec_res", 4, 128, 64, 64, 128),
          ("dec_up1", 4, 128, 128, 128, 64), ("dec_up2", 4, 64, 256, 256, 32)]
def t(fn, it=8):
    fn(); torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / it * 1e3
cands = [(bm, bn, sp) for bm in (64, 128, 256) for bn in (64, 128, 256) for sp in (1, 2, 3, 4, 6, 7, 8, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 64, 80, 96, 102, 128, 160, 204, 256)]
os.environ["SCDA_PLAN_ALLOW_BM64"] = "1"
for name, B, Cin, H, W, Cout in LAYERS:
    x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
    y = N.conv2d_fwd(x, w, b, 1, 1, 1); dy = torch.randn_like(y)
    ops = {"fwd": lambda: N.conv2d_fwd(x, w, b, 1, 1, 1), "dgrad": lambda: N.conv2d_dgrad(dy, w, x.shape, 1, 1),
           "wgrad": lambda: N.conv2d_wgrad(dy, x, w.shape, 1, 1)}
    for op, fn in ops.items():
        os.environ.pop("SCDA_PLAN_FORCE", None)
        base = t(fn)
        best = (base, "model")
        seen = set()
        for c in cands:
            os.environ["SCDA_PLAN_FORCE"] = "%d,%d,%d" % c
            try:
                v = t(fn, it=4)
            except Exception:
                continue
            key = round(v, 1)
            if v < best[0]: best = (v, c)
        os.environ.pop("SCDA_PLAN_FORCE", None)
        print("%-8s %-5s model %7.1f us   best %7.1f us %-16s gain %4.1f %%" % (name, op, base, best[0], str(best[1]), 100 * (base - best[0]) / base), flush=True)
