import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
def stat():
    d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
    return int(d["usage_usec"]), int(d["nr_throttled"]), int(d["throttled_usec"])
import numpy as np, torch
nt = os.environ.get("TORCH_THREADS")
if nt: torch.set_num_threads(int(nt))
import bench
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
src, tgt, gts, info = bench.synth_batch(0); src, tgt = src.to(dev), tgt.to(dev)
for i in range(8): tr.step(src, gts, info, tgt)
torch.cuda.synchronize()
u0, n0, t0 = stat(); w0 = time.perf_counter()
ts = []
for i in range(30):
    t = time.perf_counter(); tr.step(src, gts, info, tgt); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
w = time.perf_counter() - w0; u1, n1, t1 = stat()
print("%-40s median %.1f ms  mean %.1f  cpu-cores-busy %.1f  throttled periods %d  torch threads %d" %
      (os.environ.get("TAG", ""), np.median(ts), np.mean(ts), (u1 - u0) / 1e6 / w, n1 - n0, torch.get_num_threads()))
