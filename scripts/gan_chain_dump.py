"""Kernel-by-kernel dump of the GAN part of one steady-state iteration from a rocprofv3 kernel-trace CSV (start offset, duration,
queue, workgroups, kernel): the chain view at full resolution.   python scripts/gan_chain_dump.py <kernel_trace.csv> [from_kernel]"""
import csv, glob, re, sys
path = glob.glob(sys.argv[1])[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0); wx = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
        gy = int(r.get("Grid_Size_Y", 1) or 1); gz = int(r.get("Grid_Size_Z", 1) or 1)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), gx * gy * gz // max(wx, 1), wx))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
first, last = adam[-9] + 1, adam[-5] + 2          # the last-but-one full iteration (+ its pack kernel)
it = rows[first:last + 1]
def short(n):
    n = n.replace("scda::", "").replace("void ", "")
    return re.sub(r"\(.*", "", n)[:60]
start_key = sys.argv[2] if len(sys.argv) > 2 else "upsample2_fwd"
i0 = 0 if start_key == "all" else next(i for i, r in enumerate(it) if start_key in r[2])
# back up to the decoder's first kernel: the first conv_igemm <64,64> before the first upsample
t0 = it[0][0]
qs = sorted(set(r[3] for r in it))
print("iteration %.2f ms, %d kernels; queues %s" % ((it[-1][1] - t0) / 1e6, len(it), qs))
prev_end = {}
for s, e, n, q, wgs, wx in it[max(0, i0 - 30):]:
    gap = (s - prev_end.get(q, s)) / 1e3
    print("%8.3f ms  %7.1f us  q%-2s gap %6.1f  wg %6d x%4d  %s" % ((s - t0) / 1e6, (e - s) / 1e3, qs.index(q), gap, wgs, wx, short(n)))
    prev_end[q] = e
