"""Feasibility probe for DESIGN.md section 6 (next, 1): can torch.cuda.graph capture a module of this library -- kernels launched
over ctypes on torch's current stream, the A / B halves on two streams -- and replay it?  Captures the image discriminators'
forward (no autograd, no dropout: 2 x (3 stride-2 convs + 1x1 conv), split-K reduces, two streams) on static inputs, replays it on
new data, compares with the eager result, and times eager vs replay (host time per call and device time)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
dis = tr.dis
xa = torch.randn(4, 3, 256, 256, device=dev); xb = torch.randn(4, 3, 256, 256, device=dev)
with torch.no_grad():
    for _ in range(3):
        ra, rb = dis(xa, xb)          # warm-up: workspaces, packed weights, planner cache
torch.cuda.synchronize()
sa, sb = xa.clone(), xb.clone()       # static inputs
g = torch.cuda.CUDAGraph()
try:
    with torch.no_grad(), torch.cuda.graph(g):
        oa, ob = dis(sa, sb)
except Exception as e:  # noqa: BLE001
    print("CAPTURE FAILED:", type(e).__name__, str(e)[:400]); sys.exit(0)
ya = torch.randn(4, 3, 256, 256, device=dev); yb = torch.randn(4, 3, 256, 256, device=dev)
sa.copy_(ya); sb.copy_(yb)
g.replay(); torch.cuda.synchronize()
with torch.no_grad():
    ea, eb = dis(ya, yb)
torch.cuda.synchronize()
print("replay == eager:", torch.equal(oa, ea), torch.equal(ob, eb), " max diff", float((oa - ea).abs().max()), float((ob - eb).abs().max()))


def timeit(fn, n=200):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


def eager():
    with torch.no_grad(): dis(ya, yb)
he, de = timeit(eager); hg, dg = timeit(g.replay)
print("eager : host %.1f us / call, wall %.1f us / call" % (he, de))
print("graph : host %.1f us / call, wall %.1f us / call" % (hg, dg))
