#!/bin/bash
# round 4: cache-policy bits of the weight-gradient kernel's LDS-DMA operand stream (default " nt"), same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
for tag in default wgsc1nt wgsc0sc1nt wgsc0sc1 wgsc1 wgplain; do
  if [ "$tag" = default ]; then unset SPARF_LIB; else export SPARF_LIB=$PWD/sparf_amd/libsparf_hip_$tag.so; fi
  echo "== rep $rep lib $tag $(timeout 300 python tools/kernel_bench.py bf16x3 2>&1 | grep -E '^(wgrad|pass bwd)' | tr '\n' ' ')"
done
done > gpurun_out/r04y_wgrad_cache_policy.log 2>&1
cat gpurun_out/r04y_wgrad_cache_policy.log
