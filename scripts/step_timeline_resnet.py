import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd import _timing as T
from scda_amd import resnet_config as RC
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = RC.make_trainer(bench.CFG, dev, lr=1.25e-5)
src, tgt, gts, info = bench.synth_batch(0, RC.H, RC.W); src, tgt = src.to(dev), tgt.to(dev)
for i in range(8): tr.step(src, gts, info, tgt)
torch.cuda.synchronize()
T.ENABLED = True
rows = []
for i in range(8):
    T.MARKS.clear(); torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(src, gts, info, tgt); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    prev = t0; seg = {}
    for lab, t in T.MARKS:
        seg[lab] = (t - prev) * 1e3; prev = t
    seg['final_sync'] = (t2 - t1) * 1e3; seg['TOTAL'] = (t2 - t0) * 1e3
    rows.append(seg)
labs = [l for l, _ in T.MARKS] + ['final_sync', 'TOTAL']
for l in labs:
    print("%-26s" % l, " ".join("%5.1f" % r.get(l, 0) for r in rows))
