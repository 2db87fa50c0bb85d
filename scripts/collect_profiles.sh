#!/bin/bash
# Everything under profiles/ comes from this script, run on the MI355X box through gpurun:
#   gpurun -- 'bash scripts/collect_profiles.sh r01'
# 1. bench.py (the judged line)  2. the same command under rocprofv3 --kernel-trace --stats
# 3. + 4. PMC passes (FETCH_SIZE, WRITE_SIZE separately; never combined with other trace domains)
# 5. PMC pass for MFMA pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE)
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python $R/bench.py > $OUT/bench_1gpu.json 2> $OUT/bench_1gpu.err      # the driver's exact command (defaults: 200 steps, 5 warm-up)
export SCDA_BENCH_NO_TEMPLATE_PASS=1   # the traces below hold exactly warm-up + timed iterations
rocprofv3 --kernel-trace --stats -d $OUT/kt -o prof -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/kt.log 2>&1
grep '^{' $OUT/kt.log > $OUT/bench_under_rocprof.json
# (graphs off + the library's launch log: every Winograd dispatch of these two passes is joined with its layer by pmc_summary.py)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -f $OUT/wino_log_$c.txt
  SCDA_GAN_GRAPH=0 SCDA_WINO_LOG=$OUT/wino_log_$c.txt rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_mfma.log 2>&1
cd $R
python scripts/pmc_mfma_summary.py $OUT/pmc_mfma/pmc_counter_collection.csv $OUT/pmc_mfma.md > /dev/null
python scripts/rocprof_summary.py $(ls $OUT/kt/*.db | head -1) $OUT/kernel_stats.md "python bench.py --steps 10 --warmup 3 --no-cpu-baseline under rocprofv3 --kernel-trace --stats (13 iterations incl. warm-up)"
python scripts/pmc_summary.py $OUT $OUT/pmc_traffic.md $OUT/pmc_traffic.json
rm -rf $OUT/kt/*.db
cat $OUT/bench_1gpu.json
