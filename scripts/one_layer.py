"""run ONE conv layer / direction in a loop (for rocprofv3 counter passes): python scripts/one_layer.py conv3_2 fwd 20"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scda_amd import native
L = {"conv1_2": (1, 64, 512, 1024, 64), "conv2_2": (1, 128, 256, 512, 128), "conv3_2": (1, 256, 128, 256, 256),
     "conv4_2": (1, 512, 64, 128, 512), "conv5_x": (1, 512, 32, 64, 512), "dec_res": (4, 128, 64, 64, 128),
     "dec_up2": (4, 64, 256, 256, 32), "conv3_1": (1, 128, 128, 256, 256), "conv2_1": (1, 64, 256, 512, 128),
     "conv4_1": (1, 256, 64, 128, 512), "dec_up1": (4, 128, 128, 128, 64)}
# ResNet-50 C4 at 800 x 1344 (1x1 convolutions of the bottlenecks): name -> (B, Cin, H, W, Cout) with a "p" prefix
P = {"p1_256_64": (1, 256, 200, 336, 64), "p1_64_256": (1, 64, 200, 336, 256), "p2_512_128": (1, 512, 100, 168, 128),
     "p2_128_512": (1, 128, 100, 168, 512), "p3_1024_256": (1, 1024, 50, 84, 256), "p3_256_1024": (1, 256, 50, 84, 1024),
     "p4_2048_512": (1, 2048, 3584, 7, 512), "p4_512_2048": (1, 512, 3584, 7, 2048)}
name, what, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
FC = {"fc6": (512, 25088, 4096), "fc7": (512, 4096, 4096)}      # (RoIs, in, out): the dense GEMMs (SCDA_GEMM_X9=0: the fp32-MFMA kernel)
if name in FC:
    R, fin, fout = FC[name]
    dev = torch.device("cuda:0")
    x = torch.randn(R, fin, device=dev).clamp_min(0); w = torch.randn(fout, fin, device=dev) / fin ** 0.5; dy = torch.randn(R, fout, device=dev)
    dw = torch.zeros(fout, fin, device=dev)
    fn = {"fwd": lambda: native.linear_fwd(x, w, None), "dgrad": lambda: native.linear_dgrad(dy, w),
          "wgrad": lambda: native.linear_wgrad(dy, x, out=dw, accumulate=False)}[what]
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(name, what, native.last_plan())
    sys.exit(0)
ks, pad = (1, 0) if name in P else (3, 1)
B, Cin, H, W, Cout = (P if name in P else L)[name]
dev = torch.device("cuda:0")
x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, ks, ks, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
y = native.conv2d_fwd(x, w, b, 1, pad, 1); dy = torch.randn_like(y)
fn = {"fwd": lambda: native.conv2d_fwd(x, w, b, 1, pad, 1), "dgrad": lambda: native.conv2d_dgrad(dy, w, x.shape, 1, pad),
      "wgrad": lambda: native.conv2d_wgrad(dy, x, w.shape, 1, pad)}[what]
for _ in range(n):
    fn()
torch.cuda.synchronize()
print(name, what, native.last_plan())
