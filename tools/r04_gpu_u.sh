#!/bin/bash
# round 4: probes of the interleaved 8-bit weight-gradient kernel (wrong results): no barrier / no conversion units / no in-loop DMA
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for tag in default nobar nounits nodma nounitsdma; do
  if [ "$tag" = default ]; then unset SPARF_LIB; else export SPARF_LIB=$PWD/sparf_amd/libsparf_hip_$tag.so; fi
  echo "== lib $tag $(timeout 300 python tools/kernel_bench.py bf16+q8 2>&1 | grep -E '^wgrad')"
done > gpurun_out/r04u_wgrad_q8_probes.log 2>&1
cat gpurun_out/r04u_wgrad_q8_probes.log
