"""host CPU seconds burned per iteration (all threads of this process), optionally with HIP blocking sync"""
import ctypes, os, sys, time
mode = sys.argv[1] if len(sys.argv) > 1 else "spin"
if mode != "spin":
    hip = ctypes.CDLL("libamdhip64.so")
    flag = {"yield": 2, "block": 4}[mode]
    print("hipSetDeviceFlags ->", hip.hipSetDeviceFlags(ctypes.c_uint(flag)))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from scda_amd.train_step import ScdaTrainer
dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
src, tgt, gts, info = bench.synth_batch(0); src, tgt = src.to(dev), tgt.to(dev)
for _ in range(8): tr.step(src, gts, info, tgt)
torch.cuda.synchronize()
def threads():
    out = {}
    for t in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % t).read()
            comm = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[t] = (comm, int(rest[11]), int(rest[12]))   # utime, stime in clock ticks
        except Exception:
            pass
    return out
th0 = threads()
c0 = os.times(); t0 = time.perf_counter()
n = 60
for _ in range(n): tr.step(src, gts, info, tgt)
torch.cuda.synchronize()
c1 = os.times(); t1 = time.perf_counter()
th1 = threads()
tick = os.sysconf("SC_CLK_TCK")
rows = []
for t, (comm, u, sy) in th1.items():
    u0, s0 = th0.get(t, (comm, 0, 0))[1:]
    if (u - u0) + (sy - s0) > 0:
        rows.append(((u - u0 + sy - s0) / tick / n * 1e3, comm, (u - u0) / tick / n * 1e3, (sy - s0) / tick / n * 1e3))
for r in sorted(rows, reverse=True)[:8]:
    print("   thread %-18s %6.2f ms/iter (user %.2f sys %.2f)" % (r[1], r[0], r[2], r[3]))
cpu = (c1.user - c0.user) + (c1.system - c0.system)
print("%s: %.2f ms/iter wall, %.2f ms/iter CPU (user %.2f sys %.2f) = %.2f cores busy" % (mode, (t1 - t0) / n * 1e3, cpu / n * 1e3, (c1.user - c0.user) / n * 1e3, (c1.system - c0.system) / n * 1e3, cpu / (t1 - t0)))
