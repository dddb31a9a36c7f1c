#!/bin/bash
# PMC passes over the per-kernel micro-benchmark (one counter group per pass, as the
# MI355X guide prescribes; no sys/hip/hsa trace domains together with --pmc).
# Usage (on the GPU box): bash tools/pmc_profile.sh <tag> [bf16|fp32]
set -u
TAG=${1:-r01}
PREC=${2:-bf16}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT -o $name -- python tools/kernel_bench.py $PREC > $OUT/$name.log 2>&1
  echo "pass $name exit $?"
done
ls -la $OUT | head -30
