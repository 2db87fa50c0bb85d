"""The detector's Adam step with the classifier + heads' slice on a stream of its own (SCDA_ADAM_OVERLAP=1) against the plain step:
identical parameters after a few iterations?  iteration time?      [GPU_MAX_HW_QUEUES=8 SCDA_GAN_GRAPH=0] python scripts/adam_overlap_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0")


def run(overlap, steps, timed):
    os.environ["SCDA_ADAM_OVERLAP"] = "1" if overlap else "0"
    torch.manual_seed(0); np.random.seed(100)
    tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
    src, tgt, gts, info = bench.synth_batch(0); src, tgt = src.to(dev), tgt.to(dev)
    for _ in range(steps): tr.step(src, gts, info, tgt)
    torch.cuda.synchronize()
    snap = {k: f.data.clone() for k, f in tr.flat.items()}
    ms = None
    if timed:
        for _ in range(6): tr.step(src, gts, info, tgt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(timed): tr.step(src, gts, info, tgt)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / timed * 1e3
    return snap, ms


mode = sys.argv[1] if len(sys.argv) > 1 else "both"
if mode == "both":
    a, _ = run(False, 3, 0)
    b, _ = run(True, 3, 0)
    for k in a:
        print("bucket %-10s identical after 3 iterations: %s" % (k, torch.equal(a[k], b[k])))
else:
    _, ms = run(mode == "1", 3, 30)
    print("SCDA_ADAM_OVERLAP=%s GRAPH=%s GPU_MAX_HW_QUEUES=%s prio=%s: %.2f ms / iteration" % (mode, os.environ.get("SCDA_GAN_GRAPH", "default"),
          os.environ.get("GPU_MAX_HW_QUEUES", "default"), os.environ.get("SCDA_ADAM_PRIO", "0"), ms))
