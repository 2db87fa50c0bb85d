"""is there a periodic device-side stall?  back-to-back ~60us kernels, an event every 50 launches"""
import time, torch
x = torch.randn(32 * 1024 * 1024, device="cuda")
for _ in range(200): x.mul_(1.0001)
torch.cuda.synchronize()
evs = []
t0 = time.perf_counter()
for i in range(400):
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    for _ in range(50): x.mul_(1.0001)
e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
host_done = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
d = [evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)]
import statistics
med = statistics.median(d)
cum = 0; out = []
for v in d:
    cum += v
    if v > 1.5 * med: out.append((round(cum, 1), round(v - med, 1)))
print("median chunk %.2f ms, total %.0f ms, host enqueue done at %.0f ms, wall %.0f" % (med, sum(d), host_done * 1e3, wall * 1e3))
print("outliers (t_ms, extra_ms):", out)
