import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
if len(sys.argv) > 1 and sys.argv[1] == "noearly": tr.early_backward = False
src, tgt, gts, info = bench.synth_batch(0); src, tgt = src.to(dev), tgt.to(dev)
ts = []
for i in range(40):
    torch.cuda.synchronize(); t = time.perf_counter(); tr.step(src, gts, info, tgt); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
print(" ".join("%.0f" % t for t in ts)); print("median last 20: %.1f" % np.median(ts[20:]))
import gc
st0 = torch.cuda.memory_stats()
gc.disable()
ts = []
for i in range(24):
    torch.cuda.synchronize(); t = time.perf_counter(); tr.step(src, gts, info, tgt); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
gc.enable()
st1 = torch.cuda.memory_stats()
print("gc disabled:", " ".join("%.0f" % t for t in ts))
for k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "allocation.all.allocated", "segment.all.allocated"):
    print(k, st0.get(k), "->", st1.get(k))
print("reserved GB", torch.cuda.memory_reserved() / 2**30, "allocated GB", torch.cuda.memory_allocated() / 2**30)
