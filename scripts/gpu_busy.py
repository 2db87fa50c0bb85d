"""union-of-kernels busy time per iteration from a rocprofv3 --kernel-trace db (steady-state window)"""
import sqlite3, sys
import numpy as np
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select queue_id, start, end, name from kernels order by start"))
s = np.array([r[1] for r in rows], dtype=np.int64); e = np.array([r[2] for r in rows], dtype=np.int64); q = np.array([r[0] for r in rows])
names = [r[3] for r in rows]
# iteration boundaries: the detector's Adam launch (largest adam_kernel grid) closes an iteration
adam = [i for i, n in enumerate(names) if 'adam_kernel' in n]
ends = [e[i] for i in adam]
big = [ends[i] for i in range(3, len(ends), 4)]   # 4 adam launches per iteration, the 4th is the detector's
big = big[3:]                                      # drop warm-up
for a, b in zip(big[:-1], big[1:]):
    m = (s >= a) & (e <= b)
    ss, ee = s[m], e[m]
    o = np.argsort(ss); ss, ee = ss[o], ee[o]
    busy = 0; cs, ce = ss[0], ee[0]
    for x, y in zip(ss[1:], ee[1:]):
        if x > ce: busy += ce - cs; cs, ce = x, y
        else: ce = max(ce, y)
    busy += ce - cs
    gaps = ss[1:] - np.maximum.accumulate(ee)[:-1]
    gaps = gaps[gaps > 0]
    print("iteration %.2f ms: union busy %.2f ms (%.1f%%), kernel-sum %.2f ms, %d kernels, idle gaps: n=%d total %.2f ms, >50us: %.2f ms" %
          ((b - a) / 1e6, busy / 1e6, 100 * busy / (b - a), (ee - ss).sum() / 1e6, m.sum(), len(gaps), gaps.sum() / 1e6, gaps[gaps > 50000].sum() / 1e6))
