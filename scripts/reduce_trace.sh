#!/bin/bash
# per-launch durations of the split-K reduce kernels inside the bench (one iteration's worth, sorted by time)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/reduce_trace
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o t -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $OUT/log.txt 2>&1
cd $R
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
pat = "${1:-splitk_reduce}"
sel = [r for r in rows if pat in r["Kernel_Name"]]
n_iter = 6
agg = collections.defaultdict(list)
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    key = (r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))
    agg[key].append(d)
tot = 0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v)
    print("%-42s grid %9s wg %5s  n/iter %5.1f  avg %7.1f us  sum/iter %7.1f us" % (k[0], k[1], k[2], len(v) / n_iter, sum(v) / len(v), sum(v) / n_iter))
print("total/iter %.1f us" % (tot / n_iter))
PY
