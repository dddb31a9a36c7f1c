#!/bin/bash
# round 4: is the MFMA phase of the weight-gradient kernel bound by its LDS fragment reads?  (probe builds, wrong results: no conversion,
# no in-loop DMA, every 2nd / 4th / no streamed fragment read)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for tag in nounitsdma cfrag2 cfrag4 cfrag99; do
  export SPARF_LIB=$PWD/sparf_amd/libsparf_hip_$tag.so
  echo "== lib $tag $(timeout 300 python tools/kernel_bench.py bf16+q8 2>&1 | grep -E '^wgrad')"
done > gpurun_out/r04v_wgrad_frag_probes.log 2>&1
cat gpurun_out/r04v_wgrad_frag_probes.log
