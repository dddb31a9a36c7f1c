#!/bin/bash
# round 4, GPU session J: reference callers (4 settings x 2 modes), timing of the reference's iteration code on both renderers
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_reference_callers_gpu.py tests/test_graph_gpu.py -m gpu -q -k "callers or chunked" > gpurun_out/r04j_reference_callers.log 2>&1; echo "callers rc=$?"; tail -3 gpurun_out/r04j_reference_callers.log
timeout 900 python tests/tools/reference_callers_timing.py > gpurun_out/r04j_timing.log 2>&1; echo "timing rc=$?"; grep -E "^dtu|^llff|^replica|wrote|Error" gpurun_out/r04j_timing.log
