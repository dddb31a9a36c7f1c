#!/bin/bash
# Round-3 GPU session H: tile inputs staged through LDS (forward) -- GPU suite, same-box A/B against the previous kernels
# (sparf_amd/libsparf_hip_base.so, tools/build_variant.py HEAD~ base), SP_PROF wave-time accounting of the new forward.
set -u
TAG=r03h
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== pytest -m gpu -x"; timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1; tail -4 gpurun_out/${TAG}_pytest.log; grep -n "^E " gpurun_out/${TAG}_pytest.log | head
echo "== A/B"; bash tools/ab_kernels.sh base 2>&1 | tee gpurun_out/${TAG}_kernel_ab_staged_inputs.log
echo "== SP_PROF"; SPARF_ABI_ANY=1 SPARF_LIB=$PWD/sparf_amd/libsparf_hip_prof.so timeout 300 python tools/kernel_bench.py bf16x3 2>&1 | grep -A1 "^fwd save" | tee gpurun_out/${TAG}_wave_time_accounting.log
echo "== bench (new)"; timeout 600 python bench.py --no-psnr --no-cpu-baseline --no-other-sizes > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cut -c1-300 gpurun_out/${TAG}_bench.json
