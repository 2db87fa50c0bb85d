"""validate() at 512 x 1024 on synthetic images: time per image with the per-(class, image) NMS round trips (SCDA_NMS_UNBATCHED=1)
and with the batched per-class NMS (scda_nms_segments_hip) -- and that both write the same results."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd.train_step import ScdaTrainer
from scda_amd import evaluate

dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
src, tgt, gts, info = bench.synth_batch(0); src, tgt = src.to(dev), tgt.to(dev)
for _ in range(8):
    tr.step(src, gts, info, tgt)
torch.cuda.synchronize()
N = 24
items = []
for i in range(N):
    s, _, g, inf = bench.synth_batch(10 + i)
    items.append((s, inf, g, ["img%03d.png" % i]))
out = {}
for mode in ("unbatched", "batched", "unbatched", "batched"):
    if mode == "unbatched":
        os.environ["SCDA_NMS_UNBATCHED"] = "1"
    else:
        os.environ.pop("SCDA_NMS_UNBATCHED", None)
    d = tempfile.mkdtemp()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    evaluate.validate(items, tr.model, bench.CFG, d, score=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N * 1e3
    rows = open(os.path.join(d, "results.txt.rank0")).read()
    out.setdefault(mode, []).append((dt, rows))
    print("%-10s %.2f ms / image   (%d result rows)" % (mode, dt, rows.count("\n")), flush=True)
assert out["batched"][0][1] == out["unbatched"][0][1], "the two NMS paths wrote different results"
a, b = min(t for t, _ in out["unbatched"]), min(t for t, _ in out["batched"])
print("validate() per 512x1024 image: %.2f -> %.2f ms (x%.2f), identical results.txt" % (a, b, a / b))
