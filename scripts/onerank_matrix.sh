#!/bin/bash
# What a rank of a data-parallel run costs on ONE GPU, by process environment (scda_amd.hostenv.data_parallel_env):
#   gpurun -- 'bash scripts/onerank_matrix.sh [tag]'   -> gpurun_out/<tag>/onerank_rccl.txt  (kept as profiles/rNN_onerank_rccl.txt)
# Each line = scripts/onerank_rccl_cost.py in a fresh process: the plain step (collectives=False) or the whole RCCL choreography of
# a step on a one-rank group (collectives=True: 5 asynchronous all-reduces, their stream, events and waits; nothing is moved).
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-onerank}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; O=$OUT/onerank_rccl.txt; : > $O
run() { C=$1; shift; env "$@" python $R/scripts/onerank_rccl_cost.py $C 2>>$OUT/err.txt | grep collectives >> $O; }
run 0 A=1                                                    # single-GPU default: 4 hardware queues, GAN phases as hipGraphs
run 0 SCDA_GAN_GRAPH=0                                       # ... eager
run 0 GPU_MAX_HW_QUEUES=8 SCDA_GAN_GRAPH=0                   # eager, 8 queues
run 0 GPU_MAX_HW_QUEUES=8                                    # graphs + 8 queues: the combination that does not work
run 1 A=1                                                    # collectives, the runtime's default 4 queues
run 1 SCDA_SEGMENTED_REDUCE=0                                # ... the detector's reduction in one piece behind the backward
run 1 GPU_MAX_HW_QUEUES=6
run 1 GPU_MAX_HW_QUEUES=8                                    # what data_parallel_env sets
run 1 GPU_MAX_HW_QUEUES=8 SCDA_GAN_GRAPH=1
run 1 GPU_MAX_HW_QUEUES=8 SCDA_BLOCKING_SYNC=1               # blocking waits (hostenv.wants_blocking_sync: only under a tight CPU quota)
run 1 GPU_MAX_HW_QUEUES=8 SCDA_BLOCKING_SYNC=1 TORCH_NCCL_ASYNC_ERROR_HANDLING=0
run 1 GPU_MAX_HW_QUEUES=8 SCDA_BLOCKING_SYNC=1 TORCH_NCCL_ENABLE_MONITORING=0
run 1 GPU_MAX_HW_QUEUES=8 SCDA_BLOCKING_SYNC=1 TORCH_NCCL_ASYNC_ERROR_HANDLING=0 TORCH_NCCL_ENABLE_MONITORING=0   # (ProcessGroupNCCL's watchdog / monitor threads: within the run-to-run spread of the blocking mode)
cat $O
