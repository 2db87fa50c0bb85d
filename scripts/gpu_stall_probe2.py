"""periodic stall probe under MFMA-heavy load: back-to-back conv launches, one event per 4 launches"""
import sys, os, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scda_amd import native
dev = torch.device("cuda:0")
x = torch.randn(1, 128, 256, 512, device=dev); w = torch.randn(128, 128, 3, 3, device=dev) * 0.03; b = torch.zeros(128, device=dev)
for _ in range(20): native.conv2d_fwd(x, w, b, 1, 1, 1)
torch.cuda.synchronize()
evs = []
for i in range(500):
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    for _ in range(4): native.conv2d_fwd(x, w, b, 1, 1, 1)
e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
torch.cuda.synchronize()
d = [evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)]
med = statistics.median(d); cum = 0; out = []
for v in d:
    cum += v
    if v > 1.3 * med: out.append((round(cum, 1), round(v - med, 1)))
print("median chunk %.2f ms (4 convs), total %.0f ms" % (med, sum(d)))
print("outliers (t_ms, extra_ms):", out[:60])
