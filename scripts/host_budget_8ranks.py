"""Does the host side of EIGHT ranks fit the 16-CPU quota of a node's container?  Only one GPU is here, so seven ranks are stood in
for by processes that burn what a rank's host side burns (scripts/host_cpu_use.py: ~1.9 cores spinning -- the launching thread + the
HIP runtime's signal thread) while THIS process runs the real iteration.  Reports iteration time, cores busy in the cgroup and
throttled CFS periods for each wait mode:
    python scripts/host_budget_8ranks.py spin|block [burn_cores_per_stand_in]"""
import ctypes, multiprocessing as mp, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 else "spin"
burn = float(sys.argv[2]) if len(sys.argv) > 2 else (1.9 if mode == "spin" else 1.45)


def burner(duty, stop):
    # one thread's worth of CPU at the given duty cycle, in 2 ms slices
    while not stop.is_set():
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.002 * duty:
            pass
        rest = 0.002 * (1.0 - duty)
        if rest > 0:
            time.sleep(rest)


def stat():
    d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
    return int(d["usage_usec"]), int(d["nr_throttled"]), int(d["throttled_usec"])


if __name__ == "__main__":
    stop = mp.Event()
    procs = []
    for r in range(7):
        left = burn
        while left > 1e-6:
            d = min(1.0, left)
            p = mp.Process(target=burner, args=(d, stop), daemon=True); p.start(); procs.append(p)
            left -= d
    if mode == "block":       # what bench.py selects for more than one rank; the trainer then keeps the GAN phases eager
        from scda_amd.hostenv import prefer_blocking_sync   # (SCDA_GAN_GRAPH=1 forces the graphs on for the comparison)
        assert prefer_blocking_sync(0) == 0
    import numpy as np, torch, bench
    from scda_amd.hostenv import cpu_quota
    from scda_amd.train_step import ScdaTrainer
    dev = torch.device("cuda:0"); torch.manual_seed(0); np.random.seed(100)
    tr = ScdaTrainer(bench.CFG, dev, lr=1.25e-5, new_w=bench.W, new_h=bench.H)
    src, tgt, gts, info = bench.synth_batch(0); src, tgt = src.to(dev), tgt.to(dev)
    for _ in range(8):
        tr.step(src, gts, info, tgt)
    torch.cuda.synchronize()
    u0, n0, t0 = stat(); w0 = time.perf_counter()
    ts = []
    for _ in range(60):
        t = time.perf_counter(); tr.step(src, gts, info, tgt); ts.append((time.perf_counter() - t) * 1e3)
    torch.cuda.synchronize()
    w = time.perf_counter() - w0; u1, n1, t1 = stat()
    stop.set()
    print("%-5s graphs=%s queues=%s  7 stand-ins x %.2f cores | iteration median %.2f ms  mean %.2f  max %.1f | cgroup %.1f cores busy of a %s-CPU quota, "
          "%d throttled periods (%.0f ms)" % (mode, os.environ.get("SCDA_GAN_GRAPH", "0 (auto)" if mode == "block" else "1 (auto)"), os.environ.get("GPU_MAX_HW_QUEUES", "4 (default)"), burn, np.median(ts), np.mean(ts), max(ts),
                                           (u1 - u0) / 1e6 / w, cpu_quota(), n1 - n0, (t1 - t0) / 1e3), flush=True)
