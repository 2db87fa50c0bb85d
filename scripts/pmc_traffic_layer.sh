#!/bin/bash
# L2 memory-side traffic of ONE layer / direction:  bash scripts/pmc_traffic_layer.sh conv3_2 fwd   (FETCH_SIZE and WRITE_SIZE in separate passes)
set -u
LAYER=${1:-conv3_2}; WHAT=${2:-fwd}
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_tl; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o pmc -- python $R/scripts/one_layer.py $LAYER $WHAT 6 > $OUT/$c.log 2>&1
done
python - <<PY
import csv, glob, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(float); names = {}
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                per[r["Dispatch_Id"]] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]
    agg = collections.defaultdict(list)
    for d, v in per.items():
        agg[names[d].split("(")[0][-60:]].append(v)
    out[c] = {k: sum(v[1:]) / max(1, len(v) - 1) for k, v in agg.items() if len(v) > 1}
for k in out["FETCH_SIZE"]:
    if "wino" in k or "igemm" in k or "wgrad" in k:
        rd, wr = 2 * 1024 * out["FETCH_SIZE"][k], 1024 * out["WRITE_SIZE"].get(k, 0)
        print("$LAYER $WHAT  %-50s read %.1f MB  write %.1f MB  total %.1f MB" % (k, rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
PY
