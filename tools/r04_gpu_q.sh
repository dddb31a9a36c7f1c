#!/bin/bash
# round 4: where the 8-bit wgrad kernel's time goes (probe builds: no conversion / no MFMAs / DMA stream only), and the exactness tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_q8_saves_gpu.py -q -s > gpurun_out/r04q_q8_tests.log 2>&1; echo "q8 tests rc=$?"; grep -v "^$" gpurun_out/r04q_q8_tests.log | tail -60
for tag in default noconv nocomp dmaonly; do
  if [ "$tag" = default ]; then unset SPARF_LIB; else export SPARF_LIB=$PWD/sparf_amd/libsparf_hip_$tag.so; fi
  echo "== lib $tag"; timeout 300 python tools/kernel_bench.py bf16+q8 2>&1 | grep -E "^wgrad"
done > gpurun_out/r04q_wgrad_q8_probes.log 2>&1
cat gpurun_out/r04q_wgrad_q8_probes.log
