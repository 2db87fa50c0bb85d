import sys, os, time, threading, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
src, tgt, gts, info = bench.synth_batch(0); src, tgt = src.to(dev), tgt.to(dev)
for i in range(8): tr.step(src, gts, info, tgt)
torch.cuda.synchronize()
main_id = threading.get_ident(); samples = []; stop = False
def sampler():
    while not stop:
        fr = sys._current_frames().get(main_id)
        if fr is not None:
            st = traceback.extract_stack(fr, limit=6)
            samples.append((time.perf_counter(), " < ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(st))))
        time.sleep(0.001)
th = threading.Thread(target=sampler, daemon=True); th.start()
for i in range(40): tr.step(src, gts, info, tgt)
torch.cuda.synchronize(); stop = True; th.join()
# runs of identical innermost frames lasting > 8 ms
runs = []; i = 0
while i < len(samples):
    j = i
    while j + 1 < len(samples) and samples[j + 1][1].split(" < ")[0] == samples[i][1].split(" < ")[0]: j += 1
    d = (samples[j][0] - samples[i][0]) * 1e3
    if d > 8: runs.append((d, samples[i][1]))
    i = j + 1
agg = collections.Counter(); tot = collections.Counter()
for d, s in runs: agg[s] += 1; tot[s] += d
for s, n in agg.most_common(25): print("%3d x  total %6.0f ms  avg %5.1f  %s" % (n, tot[s], tot[s] / n, s))
gaps = [(samples[k + 1][0] - samples[k][0]) * 1e3 for k in range(len(samples) - 1)]
print("sampler gaps >8ms (GIL held by main):", [round(g) for g in gaps if g > 8][:40])
