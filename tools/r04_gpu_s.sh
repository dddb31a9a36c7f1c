#!/bin/bash
# round 4, 8-bit areas after the store-hazard fix and with the conversion interleaved into the previous tile's MFMAs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_q8_saves_gpu.py -q -s > gpurun_out/r04s_q8_tests.log 2>&1; echo "q8 tests rc=$?"; grep -n "entries differ\|by \|q8 vs plane\|gradient vs plane\|passed\|failed\|Error" gpurun_out/r04s_q8_tests.log | head -60
python tests/tools/debug_q8.py bf16 2>&1 | grep -v amdgpu.ids | head -30
AB_PRECS="bf16 bf16+q8 bf16x3 bf16x3+q8" bash tools/ab_kernels.sh > gpurun_out/r04s_q8_kernel_ab.log 2>&1; cat gpurun_out/r04s_q8_kernel_ab.log
