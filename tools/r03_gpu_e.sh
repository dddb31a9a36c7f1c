#!/bin/bash
# Round-3 GPU session E: where the bf16x3 training forward waits (SP_PROF wave-time accounting), DMA-spread / prefetch-depth
# variants, the large-pass test.
set -u
TAG=r03e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== pytest (new tests)"; timeout 900 python -m pytest tests/test_hip_gpu.py -m gpu -q -k "large_pass or backward" 2>&1 | tail -3
echo "== SP_PROF wave-time accounting (bf16x3 training forward, wave 0 of workgroup 0)"; SPARF_ABI_ANY=1 SPARF_LIB=$PWD/sparf_amd/libsparf_hip_prof.so timeout 300 python tools/kernel_bench.py bf16x3 2>&1 | grep -A1 "^fwd save" | tee gpurun_out/${TAG}_prof.log
echo "== kernel A/B: default, span12, span14, pf6"; AB_PRECS=bf16x3 SPARF_ABI_ANY=1 bash tools/ab_kernels.sh span12 span14 pf6 2>&1 | tee gpurun_out/${TAG}_ab.log
du -sh gpurun_out
