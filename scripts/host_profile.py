"""cProfile of the enqueueing thread over N iterations (the wait for device results shows up as time inside synchronize / .cpu() /
.item()): which Python-level functions the host's share of an iteration is made of.  python scripts/host_profile.py [n] [top]"""
import sys, os, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
src, tgt, gts, info = bench.synth_batch(0); src, tgt = src.to(dev), tgt.to(dev)
for i in range(10): tr.step(src, gts, info, tgt)
torch.cuda.synchronize()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
TOP = int(sys.argv[2]) if len(sys.argv) > 2 else 45
pr = cProfile.Profile()
pr.enable()
for i in range(N): tr.step(src, gts, info, tgt)
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumulative"):
    st = io.StringIO()
    pstats.Stats(pr, stream=st).strip_dirs().sort_stats(key).print_stats(TOP)
    lines = st.getvalue().splitlines()
    print("== sorted by %s, per iteration = value / %d" % (key, N))
    print("\n".join(l[:150] for l in lines[4:]))
