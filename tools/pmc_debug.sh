#!/bin/bash
# Extra PMC passes for diagnosing a kernel (wave-state and queue-full counters).
# Usage (on the GPU box): bash tools/pmc_debug.sh <tag> [bf16|fp32]
set -u
TAG=${1:-dbg}
PREC=${2:-bf16}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT -o pass$i -- python tools/kernel_bench.py $PREC > $OUT/pass$i.log 2>&1
  echo "pass $i exit $?"
done
python tools/pmc_raw.py $OUT $OUT/raw.csv > $OUT/raw.txt 2>&1
tail -5 $OUT/raw.txt
