"""Device-side duration of the iteration's segments without a profiler: an event on the compute stream at every phase mark
(scda_amd/_timing.py), differences between consecutive events, averaged over iterations.  Answers: how long is the GAN part of the
chain (from `crops+dec_fwd_enqueued` to the end) compared with the detector part?"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd import _timing as T
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
src, tgt, gts, info = bench.synth_batch(0); src, tgt = src.to(dev), tgt.to(dev)
for i in range(10): tr.step(src, gts, info, tgt)
torch.cuda.synchronize()
T.ENABLED = T.DEVICE = True
N = 12
acc = {}
order = []
tot = 0.0
t0 = time.perf_counter()
for i in range(N):
    T.MARKS.clear(); T.EVENTS.clear()
    tr.step(src, gts, info, tgt)
    end = torch.cuda.Event(enable_timing=True); end.record()
    torch.cuda.synchronize()
    ev = T.EVENTS + [("end", end)]
    for (la, a), (lb, b) in zip(ev, ev[1:]):
        acc[lb] = acc.get(lb, 0.0) + a.elapsed_time(b)
        if lb not in order: order.append(lb)
    tot += ev[0][1].elapsed_time(end)
wall = (time.perf_counter() - t0) / N * 1e3
print("segment (ends at mark)        device ms")
for l in order:
    print("%-28s %6.2f" % (l, acc[l] / N))
print("%-28s %6.2f   (wall per iteration incl. sync: %.2f ms)" % ("step_begin -> end", tot / N, wall))
