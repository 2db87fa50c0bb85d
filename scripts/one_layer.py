"""run ONE conv layer / direction in a loop (for rocprofv3 counter passes): python scripts/one_layer.py conv3_2 fwd 20"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scda_amd import native
L = {"conv1_2": (1, 64, 512, 1024, 64), "conv2_2": (1, 128, 256, 512, 128), "conv3_2": (1, 256, 128, 256, 256),
     "conv4_2": (1, 512, 64, 128, 512), "conv5_x": (1, 512, 32, 64, 512), "dec_res": (4, 128, 64, 64, 128),
     "dec_up2": (4, 64, 256, 256, 32), "conv3_1": (1, 128, 128, 256, 256), "conv2_1": (1, 64, 256, 512, 128),
     "conv4_1": (1, 256, 64, 128, 512), "dec_up1": (4, 128, 128, 128, 64)}
name, what, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
B, Cin, H, W, Cout = L[name]
dev = torch.device("cuda:0")
x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
y = native.conv2d_fwd(x, w, b, 1, 1, 1); dy = torch.randn_like(y)
fn = {"fwd": lambda: native.conv2d_fwd(x, w, b, 1, 1, 1), "dgrad": lambda: native.conv2d_dgrad(dy, w, x.shape, 1, 1),
      "wgrad": lambda: native.conv2d_wgrad(dy, x, w.shape, 1, 1)}[what]
for _ in range(n):
    fn()
torch.cuda.synchronize()
print(name, what, native.last_plan())
