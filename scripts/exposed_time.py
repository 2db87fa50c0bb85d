"""What the MFMA pipes wait for: from a rocprofv3 --kernel-trace CSV (start/end timestamps of every kernel of a bench run), the
wall time of the steady-state iterations is split into
    MFMA      at least one GEMM-class kernel (conv_igemm*, conv_wgrad*, gemm_*) is running
    exposed   no GEMM-class kernel is running but some other kernel is  -> attributed to the kernels running then
    idle      nothing is running (launch gaps, host waits)
Usage (on the GPU box):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_csv -o kt -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline
    python scripts/exposed_time.py gpurun_out/kt_csv/*kernel_trace.csv 6 > gpurun_out/exposed_time.md"""
import csv
import glob
import re
import sys

path = [p for a in sys.argv[1:-1] for p in glob.glob(a)][0]
steps = int(sys.argv[-1])
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# steady state = the last `steps` iterations; an iteration ends with the detector's adam_kernel (the 4th Adam launch of the step)
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
assert len(adam) >= 4 * (steps + 1), "trace holds fewer iterations than asked for"
first = adam[-4 * steps - 1] + 1          # first kernel after the Adam launch that ended the iteration before
last = adam[-1]
sel = rows[first:last + 1]
t0, t1 = sel[0][0], max(r[1] for r in sel)


def is_mfma(name):
    if "gemm_x9_fixup" in name:
        return False
    return ("conv_igemm" in name) or ("conv_wgrad" in name) or ("gemm_" in name) or ("conv_wino" in name)


def short(name):
    return re.sub(r"\(.*", "", name.replace("scda::", "").replace("void ", ""))[:60]


ev = []
for s, e, n in sel:
    ev.append((s, 1, n))
    ev.append((e, -1, n))
ev.sort(key=lambda x: (x[0], x[1]))
running = {}
n_mfma = 0
prev = t0
mfma_t = idle_t = exp_t = 0
blame = {}
for t, d, n in ev:
    dt = t - prev
    if dt > 0:
        if n_mfma > 0:
            mfma_t += dt
        elif running:
            exp_t += dt
            share = dt / len(running)
            for k in running:
                blame[short(k)] = blame.get(short(k), 0) + share
        else:
            idle_t += dt
    prev = t
    if d == 1:
        running[n] = running.get(n, 0) + 1
        n_mfma += is_mfma(n)
    else:
        running[n] -= 1
        if running[n] == 0:
            del running[n]
        n_mfma -= is_mfma(n)
tot = t1 - t0
print("# exposed time: %d steady-state iterations, %.3f ms each (under rocprofv3 --kernel-trace)\n" % (steps, tot / steps / 1e6))
print("| state | ms / iteration | % |\n|---|---:|---:|")
for lab, v in (("a GEMM-class (MFMA) kernel is running", mfma_t), ("only other kernels are running (exposed)", exp_t), ("nothing is running (idle)", idle_t)):
    print("| %s | %.3f | %.1f |" % (lab, v / steps / 1e6, 100.0 * v / tot))
print("\n| exposed kernel | ms / iteration |\n|---|---:|")
for k, v in sorted(blame.items(), key=lambda kv: -kv[1])[:30]:
    print("| `%s` | %.3f |" % (k, v / steps / 1e6))
# the longest idle gaps: which kernels bracket them
gaps = []
end_so_far = sel[0][1]
last_name = sel[0][2]
for s, e, n in sel[1:]:
    if s > end_so_far:
        gaps.append((s - end_so_far, short(last_name), short(n)))
    if e > end_so_far:
        end_so_far, last_name = e, n
agg = {}
for g, a, b in gaps:
    k = (a, b)
    v = agg.setdefault(k, [0, 0])
    v[0] += g
    v[1] += 1
print("\n| idle gap between ... and ... | count / iteration | ms / iteration |\n|---|---:|---:|")
for (a, b), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print("| `%s` -> `%s` | %.1f | %.3f |" % (a, b, c / steps, g / steps / 1e6))
