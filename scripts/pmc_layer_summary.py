"""summarise scripts/pmc_layer.sh: per kernel, every counter averaged over its launches (sum over instances)"""
import collections, csv, glob, os, re, sys
out = sys.argv[1]
d = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(out, "p*", "*counter_collection.csv"))):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r['Kernel_Name'].replace("void ", "").replace("scda::", ""))
        per[(k, r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
        per[(k, r['Dispatch_Id'])]['_ns'] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    for (k, _), c in per.items():
        for n, v in c.items():
            d[k][n].append(v)
lines = []
for k, c in d.items():
    if not any(x in k for x in ("igemm", "wgrad", "gemm_glds", "gemm_x9", "wino")):
        continue
    lines.append("## %s  (%d launches, %.1f us)" % (k, len(c['_ns']), sum(c['_ns']) / len(c['_ns']) / 1e3))
    for n in sorted(c):
        if n != '_ns':
            v = c[n][1:] or c[n]     # skip the first (cold) launch
            lines.append("  %-34s %16.0f" % (n, sum(v) / len(v)))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
