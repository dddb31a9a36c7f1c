#!/bin/bash
# round 4: SQ counters of the weight-gradient kernel, bf16 operands against 8-bit operands
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for prec in bf16 bf16+q8; do
  echo "=== $prec"
  PMC_FILTER=sparf::wgrad_kernel bash tools/pmc_deep.sh $prec gpurun_out/pmc_wgrad_$prec 2>&1 | grep -v "^pass\|^$"
done > gpurun_out/r04t_pmc_wgrad.log 2>&1
cat gpurun_out/r04t_pmc_wgrad.log
