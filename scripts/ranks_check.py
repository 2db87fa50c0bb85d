import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch, bench
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0")
for r in range(8):
    torch.manual_seed(0); np.random.seed(100 + r)
    tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H, world_size=1)
    src, tgt, gts, info = bench.synth_batch(r)
    src, tgt = src.to(dev), tgt.to(dev)
    for i in range(12):
        out = tr.step(src, gts, info, tgt)
    torch.cuda.synchronize()
    print("rank-%d data: loss %.4f rcnn_cls %.4f ok" % (r, float(out['loss']), float(out['rcnn_cls'])), flush=True)
    del tr
    torch.cuda.empty_cache()
