"""Are the Winograd launches repeatable?  Every variant (tile rows, persistent, split-K, fused pool, activation mask) launched many
times on two streams at once; every output must be bit-identical to the first.   python scripts/debug_wino_repeat.py [n=60]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scda_amd import native
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
SH = [("dec_res", 4, 128, 64, 64, 128), ("dec_up1", 4, 128, 128, 128, 64), ("dec_up2", 4, 64, 256, 256, 32), ("conv5", 1, 512, 16, 32, 512),
      ("conv1_2", 1, 64, 256, 512, 64), ("conv2_2", 1, 128, 128, 256, 128), ("conv3_2", 1, 256, 64, 128, 256), ("small", 2, 32, 20, 28, 40)]
s2 = torch.cuda.Stream()
for name, B, C, H, W, M in SH:
    x = torch.randn(B, C, H, W, device=dev); w = torch.randn(M, C, 3, 3, device=dev) * 0.05; b = torch.randn(M, device=dev)
    dy = torch.randn(B, M, H, W, device=dev); msk = torch.randn(B, C, H, W, device=dev)
    uf = native.conv2d_wino_pack(w, False); ud = native.conv2d_wino_pack(w, True)
    fns = {"fwd": lambda: native.conv2d_wino(x, uf, b, M, 1, 0.01), "dgrad": lambda: native.conv2d_wino(dy, ud, None, C, for_dgrad=True)}
    for k, fn in fns.items():
        ref = fn().clone()
        bad = 0
        for i in range(n):
            with torch.cuda.stream(s2):
                y2 = fn()
            y1 = fn()
            torch.cuda.synchronize()
            bad += int(not torch.equal(y1, ref)) + int(not torch.equal(y2, ref))
        print("%-8s %-6s order %s  mismatching launches %d / %d" % (name, k, native.wino_last_order() if hasattr(native, "wino_last_order") else "", bad, 2 * n), flush=True)
