import time, os
print("loadavg", os.getloadavg(), "affinity", len(os.sched_getaffinity(0)))
gaps = []; t_end = time.perf_counter() + 6; prev = time.perf_counter()
while prev < t_end:
    now = time.perf_counter()
    if now - prev > 0.002: gaps.append((now - prev) * 1e3)
    prev = now
print("gaps >2ms in 6 s of spinning:", len(gaps), ["%.1f" % g for g in sorted(gaps)[-10:]])
import torch
x = torch.randn(1024, 1024, device="cuda")
torch.cuda.synchronize()
# launch-only jitter: 20000 tiny kernels, no syncs
ts = []
for i in range(40):
    t = time.perf_counter()
    for _ in range(500): x.add_(1.0)
    ts.append((time.perf_counter() - t) * 1e3)
torch.cuda.synchronize()
print("500 tiny launches (ms):", " ".join("%.1f" % t for t in ts))
