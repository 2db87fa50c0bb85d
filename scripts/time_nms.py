import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch
from scda_amd import native
import nms_cases
dev = torch.device("cuda:0")
for n in (2000, 6000, 12000):
    rs = np.random.RandomState(n)
    x1 = rs.uniform(0, 900, n); y1 = rs.uniform(0, 400, n)
    b = np.stack([x1, y1, x1 + rs.uniform(8, 200, n), y1 + rs.uniform(8, 150, n), np.sort(rs.uniform(0, 1, n))[::-1]], 1).astype(np.float32)
    d = torch.from_numpy(b).to(dev)
    for mk in (0, 2000):
        native.nms(d, 0.7, mk); torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True); s.record()
        for _ in range(20): keep, num = native.nms(d, 0.7, mk)
        e.record(); torch.cuda.synchronize()
        print("n=%5d max_keep=%4d: %.1f us (mask + sweep), kept %d" % (n, mk, s.elapsed_time(e) / 20 * 1e3, int(num)))
