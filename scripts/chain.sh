#!/bin/bash
# kernel-by-kernel dump of the GAN part of one iteration:  gpurun -- 'bash scripts/chain.sh [tag]'
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-chain}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
export SCDA_BENCH_NO_TEMPLATE_PASS=1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $OUT/log.txt 2>&1
cd $R
python scripts/gan_chain_dump.py "$OUT/kt/*kernel_trace.csv" ${2:-upsample2_fwd} > $OUT/dump.txt 2>&1
rm -rf $OUT/kt
