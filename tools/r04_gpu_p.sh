#!/bin/bash
# round 4, 8-bit save / gradient areas (C ABI 5): exactness tests, then same-box kernel timings of the plain and the 8-bit modes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_q8_saves_gpu.py -x -q -s > gpurun_out/r04p_q8_tests.log 2>&1; echo "q8 tests rc=$?"; tail -25 gpurun_out/r04p_q8_tests.log
AB_PRECS="bf16 bf16+q8 bf16x3 bf16x3+q8" bash tools/ab_kernels.sh > gpurun_out/r04p_q8_kernel_ab.log 2>&1; cat gpurun_out/r04p_q8_kernel_ab.log
