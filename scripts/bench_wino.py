"""Winograd F(2x2,3x3) vs the direct implicit-GEMM kernel, layer by layer (HIP events): forward and data gradient.
TF/s columns are DIRECT-convolution-equivalent FLOP / time (2 * M * pixels * C * 9); the Winograd kernel's actual MFMA work is
that / 2.25."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scda_amd import native

dev = torch.device("cuda:0")
LAYERS = [  # name, B, Cin, H, W, Cout
    ("conv1_2", 1, 64, 512, 1024, 64), ("conv2_1", 1, 64, 256, 512, 128), ("conv2_2", 1, 128, 256, 512, 128),
    ("conv3_1", 1, 128, 128, 256, 256), ("conv3_2", 1, 256, 128, 256, 256), ("conv4_1", 1, 256, 64, 128, 512),
    ("conv4_2", 1, 512, 64, 128, 512), ("conv5_x", 1, 512, 32, 64, 512), ("dec_res", 4, 128, 64, 64, 128),
    ("dec_up1", 4, 128, 128, 128, 64), ("dec_up2", 4, 64, 256, 256, 32),
]


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


only = sys.argv[1:] or None
for name, B, Cin, H, W, Cout in LAYERS:
    if only and name not in only:
        continue
    x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05; b = torch.randn(Cout, device=dev)
    dy = torch.randn(B, Cout, H, W, device=dev)
    flop = 2.0 * B * H * W * Cout * Cin * 9
    os.environ["SCDA_WINOGRAD"] = "0"
    td_f = timeit(lambda: native.conv2d_fwd(x, w, b, 1, 1, 1)); td_d = timeit(lambda: native.conv2d_dgrad(dy, w, x.shape, 1, 1))
    uf = native.conv2d_wino_pack(w, False); ud = native.conv2d_wino_pack(w, True)
    tw_f = timeit(lambda: native.conv2d_wino(x, uf, b, Cout, 1, 0.01)); tw_d = timeit(lambda: native.conv2d_wino(dy, ud, None, Cin, for_dgrad=True))
    tp = timeit(lambda: native.conv2d_wino_pack(w, False, cache=False))
    wg = ""
    if native.lib().scda_conv2d_wino_wgrad_supported(B, Cin, H, W, Cout):
        os.environ["SCDA_WINOGRAD"] = "0"
        td_w = timeit(lambda: native.conv2d_wgrad_bias(dy, x, w.shape, 1, 1))
        tw_w = timeit(lambda: native.conv2d_wino_wgrad(dy, x, w.shape, want_bias=True))
        dd = native.conv2d_wgrad_bias(dy, x, w.shape, 1, 1)[0]; dwv = native.conv2d_wino_wgrad(dy, x, w.shape, want_bias=True)[0]
        wg = " | wgrad direct %7.1f us %6.1f TF  wino %7.1f us %6.1f TF  x%.2f relerr %.1e" % (
            td_w * 1e3, flop / td_w / 1e9, tw_w * 1e3, flop / tw_w / 1e9, td_w / tw_w, ((dd - dwv).abs().max() / dd.abs().max()).item())
    err = (native.conv2d_wino(x, uf, b, Cout, 1, 0.01) - native.conv2d_fwd(x, w, b, 1, 1, 1)).abs().max().item()
    print("%-8s %6.2f GFLOP | fwd direct %7.1f us %6.1f TF  wino %7.1f us %6.1f TF  x%.2f | dgrad direct %7.1f us %6.1f TF  wino %7.1f us %6.1f TF  x%.2f | pack %6.1f us | max|diff| %.2e"
          % (name, flop / 1e9, td_f * 1e3, flop / td_f / 1e9, tw_f * 1e3, flop / tw_f / 1e9, td_f / tw_f, td_d * 1e3, flop / td_d / 1e9,
             tw_d * 1e3, flop / tw_d / 1e9, td_d / tw_d, tp * 1e3, err) + wg, flush=True)
