"""the bench iteration with the images handed over as HOST buffers (pinned, uploaded asynchronously every step):
the PCIe-inclusive rate asked for next to bench.py's resident-input number"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
src, tgt, gts, info = bench.synth_batch(0)
src_h, tgt_h = src.pin_memory(), tgt.pin_memory()
def step():
    return tr.step(src_h.to(dev, non_blocking=True), gts, info, tgt_h.to(dev, non_blocking=True))
for _ in range(8): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 40
for _ in range(n): step()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("host images, pinned + async H2D each step: %.2f ms/iter = %.1f images/s (2 x %.1f MB over PCIe per iteration)" % (dt / n * 1e3, 2 * n / dt, src.numel() * 4 / 1e6))
