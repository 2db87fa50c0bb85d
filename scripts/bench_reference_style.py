"""What does the reference's OWN driver structure cost on top of the drop-in modules?  Same kernels, same modules, but the
iteration organised as tools/faster_rcnn_train_val.py organises it (torch.optim.Adam x 4 over plain parameters, one stream,
phases in program order, detector backward last) vs this repository's step (flat buckets + fused Adam, early detector backward,
side streams).  python scripts/bench_reference_style.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for style in (True, False):
    torch.manual_seed(0); np.random.seed(100)
    tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H, reference_style=style)
    src, tgt, gts, info = bench.synth_batch(0); src, tgt = src.to(dev), tgt.to(dev)
    for _ in range(8): out = tr.step(src, gts, info, tgt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): out = tr.step(src, gts, info, tgt)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-46s %6.2f images/s  %6.2f ms/iteration  loss %.4f" % ("reference-style driver on the drop-in modules" if style else "ScdaTrainer (what bench.py times)",
                                                                    2 * steps / dt, dt / steps * 1e3, float(out['loss'])), flush=True)
    del tr; torch.cuda.empty_cache()
