#!/bin/bash
# The whole committed evidence set of a round, from ONE snapshot, on the MI355X box:
#   gpurun --timeout 1500 -- 'bash scripts/collect_all.sh r02'
# then, back in the build container:  bash scripts/collect_all.sh --install r02   (copies the summaries into profiles/)
set -u
if [ "${1:-}" = "--install" ]; then
  TAG=${2:-r02}; S=gpurun_out/$TAG; D=profiles
  cp $S/bench_1gpu.json $D/${TAG}_bench_1gpu.json
  cp $S/bench_under_rocprof.json $D/${TAG}_bench_under_rocprof.json
  cp $S/kernel_stats.md $D/${TAG}_kernel_stats.md
  cp $S/kernel_categories.md $D/${TAG}_kernel_categories.md
  cp $S/pmc_mfma.md $D/${TAG}_pmc_mfma.md
  cp $S/pmc_traffic.md $D/${TAG}_pmc_traffic.md
  cp $S/pmc_traffic.json $D/${TAG}_pmc_traffic.json
  cp $S/exposed_time.md $D/${TAG}_exposed_time.md
  cp $S/shape_budget.txt $D/${TAG}_shape_budget.txt
  cp $S/conv_layers.txt $D/${TAG}_conv_layers.txt
  cp $S/resnet50/bench_under_rocprof.json $D/${TAG}_resnet50_bench_under_rocprof.json
  cp $S/resnet50/kernel_stats.md $D/${TAG}_resnet50_kernel_stats.md
  cp $S/resnet50/kernel_categories.md $D/${TAG}_resnet50_kernel_categories.md
  cp $S/resnet50_bench.json $D/${TAG}_resnet50_bench_1gpu.json
  [ -s $S/maskrcnn_bench.json ] && cp $S/maskrcnn_bench.json $D/${TAG}_maskrcnn_bench_1gpu.json
  [ -s $S/maskrcnn/kernel_categories.md ] && cp $S/maskrcnn/kernel_categories.md $D/${TAG}_maskrcnn_kernel_categories.md
  cp $S/resnet50/exposed_time.md $D/${TAG}_resnet50_exposed_time.md
  cp $S/resnet50/pmc_traffic.md $D/${TAG}_resnet50_pmc_traffic.md; cp $S/resnet50/pmc_traffic.json $D/${TAG}_resnet50_pmc_traffic.json
  cp $S/resnet50/pmc_mfma.md $D/${TAG}_resnet50_pmc_mfma.md
  for f in device_phase_times.txt host_cpu.txt resnet50_plan_search.txt plan_search.txt s2_conv_layers.txt host_budget.txt onerank_rccl.txt allreduce_contention.txt wino_layers.txt eval.txt; do
    [ -s $S/$f ] && cp $S/$f $D/${TAG}_$f
  done
  for f in bf16x9_probe.txt gemm_x9.txt pmc_fc6.txt; do
    [ -s $S/$f ] && cp $S/$f $D/${TAG}_$f
  done
  bad=0
  for f in $D/${TAG}_*; do      # an artefact that is empty or only a header line is a failed collection, not evidence
    if [ ! -s $f ] || [ "$(wc -l < $f)" -lt 2 -a "${f##*.}" != "json" ]; then echo "ERROR: $f is empty (or one line)"; bad=1; fi
  done
  exit $bad
fi
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
bash scripts/collect_profiles.sh $TAG > $OUT/collect.log 2>&1
export SCDA_BENCH_NO_TEMPLATE_PASS=1
python scripts/kernel_categories.py $OUT/kernel_stats.md 13 > $OUT/kernel_categories.md
( cd /tmp; export TMPDIR=/tmp
  rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_csv -o kt -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline > $OUT/kt_csv.log 2>&1 )
python scripts/exposed_time.py "$OUT/kt_csv/*kernel_trace.csv" 6 > $OUT/exposed_time.md
rm -rf $OUT/kt_csv
python scripts/shape_budget.py > $OUT/shape_budget.txt 2> $OUT/shape_budget.err
python scripts/bench_conv_layers.py > $OUT/conv_layers.txt 2> $OUT/conv_layers.err
python scripts/device_phase_times.py > $OUT/device_phase_times.txt 2>/dev/null
python scripts/time_s2_dgrad.py > $OUT/s2_conv_layers.txt 2>/dev/null
( python scripts/host_cpu_use.py spin; python scripts/host_cpu_use.py block ) > $OUT/host_cpu.txt 2>/dev/null
python bench.py --config resnet50 --steps 10 --warmup 4 --no-cpu-baseline > $OUT/resnet50_bench.json 2>/dev/null
python bench.py --config maskrcnn --steps 10 --warmup 4 --no-cpu-baseline > $OUT/maskrcnn_bench.json 2>/dev/null
mkdir -p $OUT/maskrcnn
bash scripts/kt.sh $TAG/maskrcnn "--config maskrcnn" > $OUT/maskrcnn/kernel_categories.md 2>/dev/null
mkdir -p $OUT/resnet50
bash scripts/kt.sh $TAG/resnet50 "--config resnet50" > $OUT/resnet50/kernel_categories.md 2>/dev/null
( cd /tmp; export TMPDIR=/tmp
  rocprofv3 --kernel-trace --output-format csv -d $OUT/kt_csv -o kt -- python $R/bench.py --config resnet50 --steps 6 --warmup 3 --no-cpu-baseline > $OUT/kt_csv.log 2>&1 )
python scripts/exposed_time.py "$OUT/kt_csv/*kernel_trace.csv" 6 > $OUT/resnet50/exposed_time.md
rm -rf $OUT/kt_csv
# counter passes for the ResNet-50 C4 configuration (FETCH_SIZE / WRITE_SIZE separately, then the MFMA pipe): PMC only + kernel trace
( cd /tmp; export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/resnet50/pmc_$c -o pmc -- python $R/bench.py --config resnet50 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/resnet50/pmc_$c.log 2>&1
  done
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/resnet50/pmc_mfma -o pmc -- python $R/bench.py --config resnet50 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/resnet50/pmc_mfma.log 2>&1 )
python scripts/pmc_summary.py $OUT/resnet50 $OUT/resnet50/pmc_traffic.md $OUT/resnet50/pmc_traffic.json resnet > /dev/null
python scripts/pmc_mfma_summary.py $OUT/resnet50/pmc_mfma/pmc_counter_collection.csv $OUT/resnet50/pmc_mfma.md > /dev/null
python scripts/host_budget_8ranks.py spin > $OUT/host_budget.txt 2>/dev/null
python scripts/host_budget_8ranks.py block >> $OUT/host_budget.txt 2>/dev/null
SCDA_GAN_GRAPH=1 python scripts/host_budget_8ranks.py block >> $OUT/host_budget.txt 2>/dev/null
# the process environment of a data-parallel rank (hostenv.data_parallel_env): eight hardware queues, eager GAN phases
GPU_MAX_HW_QUEUES=8 SCDA_GAN_GRAPH=0 python scripts/host_budget_8ranks.py spin >> $OUT/host_budget.txt 2>/dev/null
GPU_MAX_HW_QUEUES=8 SCDA_GAN_GRAPH=0 python scripts/host_budget_8ranks.py block >> $OUT/host_budget.txt 2>/dev/null
bash scripts/onerank_matrix.sh $TAG/onerank > /dev/null 2>&1; cp $OUT/onerank/onerank_rccl.txt $OUT/onerank_rccl.txt
[ -s scripts/micro/build/libring_standin.so ] || echo "ERROR: scripts/micro/build/libring_standin.so is missing (python -c 'import __graft_entry__ as g; g.build()' builds it in the container)" >&2
( python scripts/allreduce_contention.py 32 4; echo; echo "GPU_MAX_HW_QUEUES=8:"; GPU_MAX_HW_QUEUES=8 python scripts/allreduce_contention.py 32 4 ) 2>/dev/null > $OUT/allreduce_contention.txt
python scripts/bench_wino.py > $OUT/wino_layers.txt 2>/dev/null
# the exact-product bf16 x 9 GEMM: what the instruction does with its sums / the FC products on both kernels / counters of FC6
[ -x scripts/micro/bf16x9_probe ] && ./scripts/micro/bf16x9_probe > $OUT/bf16x9_probe.txt 2>&1
python scripts/time_gemm_x9.py 2>/dev/null | grep -v amdgpu.ids > $OUT/gemm_x9.txt
( for w in fwd dgrad wgrad; do bash scripts/pmc_layer.sh fc6 $w $TAG/pmc_fc6_$w 2>/dev/null | grep -A30 "gemm_x9_kernel"; done
  SCDA_GEMM_X9=0 bash scripts/pmc_layer.sh fc6 fwd $TAG/pmc_fc6_fwd_f32 2>/dev/null | grep -A30 "gemm_glds_kernel" ) > $OUT/pmc_fc6.txt 2>/dev/null
rm -rf $OUT/pmc_fc6_*/p[123]
python scripts/time_eval.py 2>/dev/null > $OUT/eval.txt
if [ "${2:-}" = "full" ]; then
  python scripts/tune_plans.py resnet > $OUT/resnet50_plan_search.txt 2>/dev/null
  python scripts/tune_plans.py > $OUT/plan_search.txt 2>/dev/null
fi
cat $OUT/bench_1gpu.json
