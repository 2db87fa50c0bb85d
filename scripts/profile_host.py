"""cProfile of the training iteration's host side (where does the wall time go when the GPU is idle?)"""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0")
torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
src, tgt, gts, info = bench.synth_batch(0)
src, tgt = src.to(dev), tgt.to(dev)
for _ in range(3): tr.step(src, gts, info, tgt)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): tr.step(src, gts, info, tgt)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps.print_stats(45); print(s.getvalue()[:9000])
