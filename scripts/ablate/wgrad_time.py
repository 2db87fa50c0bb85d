"""time of the Winograd weight gradient (kernel + its split-K reduce) per layer, for the ablation libraries of wino_ablate.sh:
   SCDA_OPS_LIB=scripts/ablate/build/libscda_ops_A<n>.so python scripts/ablate/wgrad_time.py [layer ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scda_amd import native

dev = torch.device("cuda:0")
LAYERS = [("conv1_2", 1, 64, 512, 1024, 64), ("conv2_2", 1, 128, 256, 512, 128), ("conv3_2", 1, 256, 128, 256, 256),
          ("conv4_2", 1, 512, 64, 128, 512), ("conv5_x", 1, 512, 32, 64, 512), ("dec_res", 4, 128, 64, 64, 128)]
only = sys.argv[1:] or None
out = []
for name, B, Cin, H, W, Cout in LAYERS:
    if only and name not in only:
        continue
    x = torch.randn(B, Cin, H, W, device=dev); dy = torch.randn(B, Cout, H, W, device=dev)
    fn = lambda: native.conv2d_wino_wgrad(dy, x, (Cout, Cin, 3, 3), want_bias=True)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 20)
    out.append("%s %.1f" % (name, best * 1e3))
print(os.path.basename(os.environ.get("SCDA_OPS_LIB", "HEAD")), " | ".join(out), flush=True)
