"""time of the Winograd forward per layer, for the ablation libraries of wino_ablate.sh / SCDA_WINO_DBG:
   SCDA_OPS_LIB=scripts/ablate/build/libscda_ops_A<n>.so [SCDA_WINO_DBG=1|2] python scripts/ablate/fwd_time.py [layer ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scda_amd import native

dev = torch.device("cuda:0")
LAYERS = [("conv4_2", 1, 512, 64, 128, 512), ("conv5_x", 1, 512, 32, 64, 512), ("dec_res", 4, 128, 64, 64, 128), ("dec_up1", 4, 128, 128, 128, 64),
          ("dec_up2", 4, 64, 256, 256, 32)]
only = sys.argv[1:] or None
out = []
for name, B, Cin, H, W, Cout in LAYERS:
    if only and name not in only:
        continue
    x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
    u = native.conv2d_wino_pack(w, False)
    fn = lambda: native.conv2d_wino(x, u, b, Cout, 1, 0.01)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 30)
    out.append("%s %.1f" % (name, best * 1e3))
print(os.path.basename(os.environ.get("SCDA_OPS_LIB", "HEAD")), "dbg=" + os.environ.get("SCDA_WINO_DBG", "0"), " | ".join(out), flush=True)
