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
# (name, batch, Cin, H, W, Cout, k, stride, pad, row_period)  -- `python scripts/tune_plans.py resnet`: the ResNet-50 C4 shapes at 800 x 1344
RESNET = [("l2_c1", 1, 512, 100, 168, 128, 1, 1, 0, 0), ("l2_c2", 1, 128, 100, 168, 128, 3, 1, 1, 0), ("l2_c3", 1, 128, 100, 168, 512, 1, 1, 0, 0),
          ("l3_c1", 1, 1024, 50, 84, 256, 1, 1, 0, 0), ("l3_c2", 1, 256, 50, 84, 256, 3, 1, 1, 0), ("l3_c3", 1, 256, 50, 84, 1024, 1, 1, 0, 0),
          ("rpn", 1, 1024, 50, 84, 512, 3, 1, 1, 0),
          ("h_c1a", 1, 1024, 3584, 7, 512, 1, 1, 0, 0), ("h_c2", 1, 512, 3584, 7, 512, 3, 1, 1, 7), ("h_c3", 1, 512, 3584, 7, 2048, 1, 1, 0, 0),
          ("h_c1b", 1, 2048, 3584, 7, 512, 1, 1, 0, 0), ("h_ds", 1, 1024, 3584, 7, 2048, 1, 1, 0, 0)]
if len(sys.argv) > 1 and sys.argv[1] == "resnet":
    LAYERS = RESNET
else:
    LAYERS = [l + (3, 1, 1, 0) for l in LAYERS]


def t(fn, it=8):
    fn(); torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / it * 1e3
cands = [(bm, bn, sp) for bm in (64, 128, 256) for bn in (64, 128, 256) for sp in (1, 2, 3, 4, 6, 7, 8, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 64, 80, 96, 102, 128, 160, 204, 256)]
os.environ["SCDA_PLAN_ALLOW_BM64"] = "1"
for name, B, Cin, H, W, Cout, k, st, pd, rp in LAYERS:
    x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
    y = N.conv2d_fwd(x, w, b, st, pd, 1, row_period=rp); dy = torch.randn_like(y)
    ops = {"fwd": lambda: N.conv2d_fwd(x, w, b, st, pd, 1, row_period=rp), "dgrad": lambda: N.conv2d_dgrad(dy, w, x.shape, st, pd, row_period=rp),
           "wgrad": lambda: N.conv2d_wgrad(dy, x, w.shape, st, pd, row_period=rp)}
    for op, fn in ops.items():
        os.environ.pop("SCDA_PLAN_FORCE", None)
        base = t(fn)
        model_plan = N.last_plan()[:3]
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
        fl = 2.0 * Cout * Cin * k * k * y.shape[0] * y.shape[2] * y.shape[3]
        print("%-8s %-5s model %7.1f us (%5.1f TF/s, plan %s)   best %7.1f us %-16s gain %4.1f %%" % (name, op, base, fl / base / 1e6, model_plan, best[0], str(best[1]), 100 * (base - best[0]) / base), flush=True)
