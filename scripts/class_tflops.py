"""achieved TFLOP/s per GEMM-class kernel inside the iteration (in-library HIP-event profiler, all classes on)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd import native
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
src, tgt, gts, info = bench.synth_batch(0); src, tgt = src.to(dev), tgt.to(dev)
for _ in range(6): tr.step(src, gts, info, tgt)
torch.cuda.synchronize()
native.prof_enable(True)
n = 10
for _ in range(n): tr.step(src, gts, info, tgt)
torch.cuda.synchronize()
native.prof_enable(False)
p = native.prof_collect()
tot_ms = tot_fl = 0
for k, (c, ms, fl, by) in sorted(p.items(), key=lambda kv: -kv[1][1]):
    print("%-44s %5.1f launches/iter %7.3f ms/iter %7.1f GFLOP/iter %6.1f TFLOP/s" % (k, c / n, ms / n, fl / n / 1e9, fl / ms / 1e9))
    tot_ms += ms; tot_fl += fl
print("TOTAL %.2f ms/iter, %.3f TFLOP/iter, %.1f TFLOP/s average" % (tot_ms / n, tot_fl / n / 1e12, tot_fl / tot_ms / 1e9))
