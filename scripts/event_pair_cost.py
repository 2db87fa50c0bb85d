"""What do the in-library profiler's HIP event pairs around the dominant kernel cost the timed region?  One process, alternating blocks
of iterations with the pairs on (bench.py's timed region) and off, so that box-to-box and thermal drift cancel."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd import native
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
src, tgt, gts, info = bench.synth_batch(0); src, tgt = src.to(dev), tgt.to(dev)
for i in range(25): tr.step(src, gts, info, tgt)
torch.cuda.synchronize()
dom = list(bench.DOMINANT_WINO)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
res = {True: [], False: []}
for blk in range(12):
    on = blk % 2 == 0
    native.prof_enable(dom if on else False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(N): tr.step(src, gts, info, tgt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N * 1e3
    native.prof_enable(False); native.prof_collect()
    res[on].append(dt)
    print("block %2d pairs %-3s %.3f ms / iteration" % (blk, "on" if on else "off", dt))
a, b = np.array(res[True]), np.array(res[False])
print("pairs on  : mean %.3f  median %.3f ms" % (a.mean(), np.median(a)))
print("pairs off : mean %.3f  median %.3f ms" % (b.mean(), np.median(b)))
print("cost of the pairs: %.3f ms / iteration (%.2f %%)" % (a.mean() - b.mean(), 100 * (a.mean() - b.mean()) / b.mean()))
