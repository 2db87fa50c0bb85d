import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
src, tgt, gts, info = bench.synth_batch(0); src, tgt = src.to(dev), tgt.to(dev)
for i in range(8): tr.step(src, gts, info, tgt)
torch.cuda.synchronize()
iv = []
for i in range(80):
    t = time.perf_counter(); tr.step(src, gts, info, tgt); torch.cuda.synchronize(); iv.append((t, time.perf_counter()))
iv = np.array(iv); iv -= iv[0, 0]; dur = (iv[:, 1] - iv[:, 0]) * 1e3
slow = dur > np.median(dur) * 1.25
print("median %.1f slow frac %.2f" % (np.median(dur), slow.mean()))
for period in (0.05, 0.1, 0.2, 0.25, 0.5, 1.0):
    best = 0
    for phi in np.linspace(0, period, 200, endpoint=False):
        k0 = np.ceil((iv[:, 0] - phi) / period); has = (k0 * period + phi) < iv[:, 1]
        agree = (has == slow).mean(); best = max(best, agree)
    print("period %.0f ms: best agreement of 'contains a tick' with 'slow' = %.2f" % (period * 1e3, best))
print(" ".join("%.0f" % d for d in dur))
