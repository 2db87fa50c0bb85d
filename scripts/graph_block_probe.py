"""host-side time per segment of the iteration in steady state (no per-iteration sync), spinning or blocking waits:
   SCDA_GAN_GRAPH=0|1 python scripts/graph_block_probe.py spin|block"""
import ctypes, os, sys, time
mode = sys.argv[1] if len(sys.argv) > 1 else "spin"
if mode != "spin":
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags ->", hip.hipSetDeviceFlags(ctypes.c_uint({"yield": 2, "block": 4}[mode])))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd import _timing as T
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
src, tgt, gts, info = bench.synth_batch(0); src, tgt = src.to(dev), tgt.to(dev)
for i in range(10): tr.step(src, gts, info, tgt)
torch.cuda.synchronize()
T.ENABLED = True
n = 40
acc, order = {}, []
t_begin = time.perf_counter()
for i in range(n):
    T.MARKS.clear(); t0 = time.perf_counter()
    tr.step(src, gts, info, tgt)
    prev = t0
    for lab, t in T.MARKS:
        if lab not in acc: acc[lab] = 0.0; order.append(lab)
        acc[lab] += (t - prev) * 1e3; prev = t
    acc['(return)'] = acc.get('(return)', 0.0) + (time.perf_counter() - prev) * 1e3
torch.cuda.synchronize()
wall = (time.perf_counter() - t_begin) / n * 1e3
print("%s SCDA_GAN_GRAPH=%s: %.2f ms/iter wall" % (mode, os.environ.get("SCDA_GAN_GRAPH", "0"), wall))
for lab in order + ['(return)']:
    print("  %-28s %6.2f ms host" % (lab, acc[lab] / n))
