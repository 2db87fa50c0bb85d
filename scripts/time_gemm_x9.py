"""FC6 / FC7's three products (forward, data gradient, weight gradient; 512 RoIs) on the fp32-MFMA kernel and on the exact-product
bf16 x 9 kernel (csrc/conv_gemm.hip gemm_x9_kernel): time per launch (HIP events, 20 launches), fp32-equivalent TFLOP/s, and the error
of each against an fp64 product on a 128 x 128 block of the result.  `x9only`: FC6 on the bf16 x 9 kernel only (scripts/ablate/x9_ablate.sh
times its ablation builds through SCDA_OPS_LIB)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from scda_amd import native
X9ONLY = len(sys.argv) > 1 and sys.argv[1] == "x9only"
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
R = 512


def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, fin, fout in ((("FC6", 25088, 4096),) if X9ONLY else (("FC6", 25088, 4096), ("FC7", 4096, 4096))):
    x = torch.randn(R, fin, generator=g).clamp_min(0).to(dev)
    w = (torch.randn(fout, fin, generator=g) / fin ** 0.5).to(dev)
    dy = (torch.randn(R, fout, generator=g) / 100).to(dev)
    dw = torch.zeros(fout, fin, device=dev)
    flop = 2.0 * R * fin * fout
    xs, ws, dys = x[:128].cpu().double(), w[:128].cpu().double(), dy.cpu().double()
    refs = {"fwd": (xs @ ws.t(), xs.abs() @ ws.abs().t()),
            "dgrad": (dys[:128] @ w[:, :128].cpu().double(), dys[:128].abs() @ w[:, :128].cpu().double().abs()),
            "wgrad": (dys[:, :128].t() @ x[:, :128].cpu().double(), dys[:, :128].abs().t() @ x[:, :128].cpu().double().abs())}
    for mode in (("1",) if X9ONLY else ("0", "1")):
        os.environ["SCDA_GEMM_X9"] = mode
        outs = {"fwd": native.linear_fwd(x, w, None), "dgrad": native.linear_dgrad(dy, w), "wgrad": native.linear_wgrad(dy, x, out=dw, accumulate=False)}
        plans = {}
        for k, fn in (("fwd", lambda: native.linear_fwd(x, w, None)), ("dgrad", lambda: native.linear_dgrad(dy, w)),
                      ("wgrad", lambda: native.linear_wgrad(dy, x, out=dw, accumulate=False))):
            ms = timed(fn)
            plan = native.last_plan()
            ref, sc = refs[k]
            e = ((outs[k][:128, :128].cpu().double() - ref).abs() / sc)
            print("%s %-5s %s  %.3f ms  %6.1f TFLOP/s (fp32-equivalent)  plan %s  error max %.2e rms %.2e of sum|ab|"
                  % (name, k, "bf16x9" if plan[3] == 2 else "fp32  ", ms, flop / ms / 1e9, plan, e.max(), e.pow(2).mean().sqrt()), flush=True)
