#!/bin/bash
# kernel trace of the bench only:  bash scripts/kt.sh <tag>   -> gpurun_out/<tag>/kernel_stats.md (+ category table)
set -u
TAG=${1:-kt}
EXTRA=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o prof -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline $EXTRA > $OUT/kt.log 2>&1
grep '^{' $OUT/kt.log > $OUT/bench_under_rocprof.json
cd $R
# iterations in the trace: 3 warm-up + 10 timed; the ResNet configuration adds 3 (its dominant class is timed in a separate pass)
ITERS=13; case "$EXTRA" in *resnet50*|*maskrcnn*) ITERS=16;; esac
python scripts/rocprof_summary.py $(ls $OUT/kt/*.db | head -1) $OUT/kernel_stats.md "python bench.py --steps 10 --warmup 3 --no-cpu-baseline $EXTRA under rocprofv3 --kernel-trace --stats ($ITERS iterations incl. warm-up)" > /dev/null
rm -rf $OUT/kt/*.db
python scripts/kernel_categories.py $OUT/kernel_stats.md $ITERS
