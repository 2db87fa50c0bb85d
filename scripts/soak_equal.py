"""N iterations with changing inputs, then a checksum of every parameter and Adam moment of the four nets -- run it under two settings
(e.g. default and SCDA_NO_NORM_UP_FUSION=1, or SCDA_GAN_GRAPH=0) and compare the lines: bit-identical training or not.
   python scripts/soak_equal.py [n]"""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
last = None
for i in range(n):
    src, tgt, gts, info = bench.synth_batch(i % 7)
    last = tr.step(src.to(dev), gts, info, tgt.to(dev))
torch.cuda.synchronize()
h = hashlib.sha256()
for name, opt in sorted(tr.opt.items()):
    for t in (opt.flat.data, opt.exp_avg, opt.exp_avg_sq):
        h.update(t.detach().cpu().numpy().tobytes())
knobs = {k: v for k, v in os.environ.items() if k.startswith("SCDA_")}
print("%d iterations  loss %.6f recon %.6f  sha256(params + Adam moments of 4 nets) %s  %s" % (n, float(last['loss']), float(last['recon_loss']), h.hexdigest()[:32], knobs), flush=True)
