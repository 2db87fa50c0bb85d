"""One steady-state iteration as a flat list: every kernel of the LAST iteration in a rocprofv3 --kernel-trace CSV with its start
(us since the iteration's first kernel), duration, queue and the idle time in front of it (time during which NOTHING ran).
Used to find what the device waits for inside an iteration (scripts/exposed_time.py gives the totals, this the places).
Usage (on the GPU box):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_csv -o kt -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline
    python scripts/iteration_kernels.py gpurun_out/kt_csv/*kernel_trace.csv > gpurun_out/iteration_kernels.txt"""
import csv
import glob
import re
import sys

path = [p for a in sys.argv[1:] for p in glob.glob(a)][0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
first, last = adam[-5] + 1, adam[-1]
sel = rows[first:last + 1]
t0 = sel[0][0]


def short(name):
    return re.sub(r"\(.*", "", name.replace("scda::", "").replace("void ", ""))[:70]


queues = {}
end_so_far = t0
print("%9s %8s %7s  q  kernel" % ("start us", "dur us", "idle us"))
for s, e, n, q, st in sel:
    qi = queues.setdefault((q, st), len(queues))
    idle = max(0, s - end_so_far)
    print("%9.1f %8.1f %7s %2d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, ("%.1f" % (idle / 1e3)) if idle > 3000 else "", qi, short(n)))
    end_so_far = max(end_so_far, e)
print("# iteration %.3f ms, %d kernels, queues %s" % ((end_so_far - t0) / 1e6, len(sel), queues))
