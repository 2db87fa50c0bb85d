"""200 iterations with changing inputs: losses stay finite, device memory and the host-side caches stop growing"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd import native
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mem = []
t0 = time.time()
for i in range(n):
    src, tgt, gts, info = bench.synth_batch(i % 7)
    out = tr.step(src.to(dev), gts, info, tgt.to(dev))
    if i % 20 == 0 or i == n - 1:
        l = float(out['loss'])
        assert np.isfinite(l), (i, l)
        torch.cuda.synchronize()
        mem.append((i, torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20, len(native._PACK_CACHE)))
        print("iter %4d loss %.4f rcnn_cls %.4f  allocated %d MiB reserved %d MiB pack-cache %d" % (i, l, float(out['rcnn_cls']), *mem[-1][1:]), flush=True)
print("%.1f s, %.1f ms/iter incl. uploads" % (time.time() - t0, (time.time() - t0) / n * 1e3))
assert mem[-1][1] <= mem[2][1] * 1.05 + 64, "allocated memory keeps growing"
assert mem[-1][3] <= mem[2][3] + 8, "pack cache keeps growing"
print("OK")
