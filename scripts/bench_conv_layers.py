"""Per-layer micro-benchmark of the MFMA conv/GEMM kernels at config-2 shapes (HIP events)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scda_amd import native

dev = torch.device("cuda:0")
LAYERS = [  # name, B, Cin, H, W, Cout, k, s, p
    ("conv1_1", 1, 3, 512, 1024, 64, 3, 1, 1), ("conv1_2", 1, 64, 512, 1024, 64, 3, 1, 1),
    ("conv2_1", 1, 64, 256, 512, 128, 3, 1, 1), ("conv2_2", 1, 128, 256, 512, 128, 3, 1, 1),
    ("conv3_1", 1, 128, 128, 256, 256, 3, 1, 1), ("conv3_2", 1, 256, 128, 256, 256, 3, 1, 1),
    ("conv4_1", 1, 256, 64, 128, 512, 3, 1, 1), ("conv4_2", 1, 512, 64, 128, 512, 3, 1, 1),
    ("conv5_x", 1, 512, 32, 64, 512, 3, 1, 1),
    ("dec_res", 4, 128, 64, 64, 128, 3, 1, 1), ("dec_up1", 4, 128, 128, 128, 64, 3, 1, 1), ("dec_up2", 4, 64, 256, 256, 32, 3, 1, 1),
]

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

rows = []
for name, B, Cin, H, W, Cout, k, s, p in LAYERS:
    x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
    b = torch.randn(Cout, device=dev)
    y = native.conv2d_fwd(x, w, b, s, p, 1)
    dy = torch.randn_like(y)
    flop = 2.0 * y.numel() * Cin * k * k
    tf = timeit(lambda: native.conv2d_fwd(x, w, b, s, p, 1))
    td = timeit(lambda: native.conv2d_dgrad(dy, w, x.shape, s, p))
    tw = timeit(lambda: native.conv2d_wgrad(dy, x, w.shape, s, p))
    rows.append((name, flop / 1e9, tf, flop / tf / 1e9, td, flop / td / 1e9, tw, flop / tw / 1e9))
    print("%-8s %7.2f GFLOP  fwd %7.3f ms %6.1f TF | dgrad %7.3f ms %6.1f TF | wgrad %7.3f ms %6.1f TF" % rows[-1], flush=True)
for (M, N, K, nm) in [(512, 4096, 25088, "fc6"), (512, 4096, 4096, "fc7")]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.01; b = torch.randn(N, device=dev); dy = torch.randn(M, N, device=dev)
    flop = 2.0 * M * N * K
    tf = timeit(lambda: native.linear_fwd(x, w, b, 1)); td = timeit(lambda: native.linear_dgrad(dy, w)); tw = timeit(lambda: native.linear_wgrad(dy, x))
    print("%-8s %7.2f GFLOP  fwd %7.3f ms %6.1f TF | dgrad %7.3f ms %6.1f TF | wgrad %7.3f ms %6.1f TF" % (nm, flop / 1e9, tf, flop / tf / 1e9, td, flop / td / 1e9, tw, flop / tw / 1e9), flush=True)
