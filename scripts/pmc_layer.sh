#!/bin/bash
# counter passes over ONE layer/direction in a loop:  bash scripts/pmc_layer.sh conv3_2 fwd [tag]
# (PMC passes only, each with --kernel-trace; never combined with other trace domains)
set -u
LAYER=${1:-conv3_2}; WHAT=${2:-fwd}; TAG=${3:-pmc_layer}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" \
  "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAVES" \
  ; do   # (the TA / TD / TCP / TCC sets of round 2 are gone: under ROCm 7.2 on these boxes the TA pass aborts rocprofv3 and then hangs until the timeout)
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $R/scripts/one_layer.py $LAYER $WHAT 6 > $OUT/p$i.log 2>&1
done
cd $R
python scripts/pmc_layer_summary.py $OUT
