"""Coarse device timeline of one steady-state iteration from a rocprofv3 kernel-trace CSV: when (ms after the iteration's first
kernel) do the landmark kernels run, and how busy is the device (union of kernel intervals) between consecutive landmarks?
    python scripts/iteration_timeline.py <kernel_trace.csv>"""
import csv, glob, re, sys
path = glob.glob(sys.argv[1])[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "?"), r.get("Queue_Id", "?")))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
first, last = adam[-9] + 1, adam[-5]          # the last-but-one full iteration
it = rows[first:last + 1]
t0 = it[0][0]
def short(n): return re.sub(r"\(.*", "", n.replace("scda::", "").replace("void ", ""))[:48]
marks = []
seen = {}
for s, e, n, st, q in it:
    k = short(n)
    for key in ("nms_sweep", "roi_pool_fwd", "roi_pool_bwd", "adam_kernel", "anchor_label", "softmax_ce_fwd", "upsample2_fwd", "instnorm_fwd_kernel<16>", "sigmoid_bce_rows_fwd", "pack_weights_batched"):
        if key in k:
            seen[key] = seen.get(key, 0) + 1
            marks.append((s, e, "%s #%d" % (key, seen[key]), q))
print("iteration: %.2f ms, %d kernels" % ((max(r[1] for r in it) - t0) / 1e6, len(it)))
# busy union between marks
ev = sorted(set([t0] + [m[0] for m in marks] + [max(r[1] for r in it)]))
def busy(a, b):
    tot = 0; cur_s = cur_e = None
    for s, e, *_ in it:
        s2, e2 = max(s, a), min(e, b)
        if e2 <= s2: continue
        if cur_e is None or s2 > cur_e:
            if cur_e is not None: tot += cur_e - cur_s
            cur_s, cur_e = s2, e2
        else:
            cur_e = max(cur_e, e2)
    if cur_e is not None: tot += cur_e - cur_s
    return tot
prev = t0
for s, e, lab, q in sorted(marks):
    seg = s - prev
    print("%7.2f ms  %-28s queue %s   [since previous mark: %5.2f ms, device busy %5.1f %%]" % ((s - t0) / 1e6, lab, q, seg / 1e6, 100.0 * busy(prev, s) / seg if seg > 0 else 0))
    prev = s
queues = {}
for s, e, n, st, q in it:
    a = queues.setdefault(q, [0, 0]); a[0] += 1; a[1] += e - s
print("per queue:", {q: (c, round(t / 1e6, 2)) for q, (c, t) in queues.items()})
