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
    if 'GRBM_GUI_ACTIVE' in c and 'SQ_VALU_MFMA_BUSY_CYCLES' in c:      # (both summed over their instances: 8 XCDs / 1024 SIMDs)
        g = sum(c['GRBM_GUI_ACTIVE'][1:] or c['GRBM_GUI_ACTIVE']) / len(c['GRBM_GUI_ACTIVE'][1:] or c['GRBM_GUI_ACTIVE']) / 8
        m = sum(c['SQ_VALU_MFMA_BUSY_CYCLES'][1:] or c['SQ_VALU_MFMA_BUSY_CYCLES']) / len(c['SQ_VALU_MFMA_BUSY_CYCLES'][1:] or c['SQ_VALU_MFMA_BUSY_CYCLES']) / 1024
        ns = sum(c['_ns'][1:] or c['_ns']) / len(c['_ns'][1:] or c['_ns'])
        lines.append("  => active cycles per XCD %.0f = %.2f GHz over the launch; MFMA pipe busy %.1f %% of them" % (g, g / ns, 100.0 * m / g))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
