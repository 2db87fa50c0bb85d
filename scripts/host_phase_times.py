"""Host-side duration of the iteration's segments (scda_amd/_timing.py marks, host clock only -- no events, no per-iteration sync:
the pipeline runs as in bench.py): where the enqueueing thread spends its iteration, waits for device results included."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd import _timing as T
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
batches = [bench.synth_batch(i) for i in range(4)]
batches = [(s.to(dev), t.to(dev), g, i) for s, t, g, i in batches]
for i in range(10):
    s, t, g, inf = batches[i % 4]; tr.step(s, g, inf, t)
torch.cuda.synchronize()
T.ENABLED = True; T.DEVICE = False
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
acc, order = {}, []
t0 = time.perf_counter()
prev_end = t0
for i in range(N):
    T.MARKS.clear()
    s, t, g, inf = batches[i % 4]
    tr.step(s, g, inf, t)
    now = time.perf_counter()
    ms = [("iteration_start", prev_end)] + list(T.MARKS) + [("step_returned", now)]
    for (la, a), (lb, b) in zip(ms, ms[1:]):
        acc[lb] = acc.get(lb, 0.0) + (b - a) * 1e3
        if lb not in order: order.append(lb)
    prev_end = now
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N * 1e3
print("segment (ends at mark)        host ms")
for l in order:
    print("%-28s %6.2f" % (l, acc[l] / N))
print("%-28s %6.2f   (wall per iteration: %.2f ms)" % ("sum", sum(acc.values()) / N, wall))
