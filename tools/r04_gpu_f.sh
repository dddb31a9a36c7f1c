#!/bin/bash
# round 4, GPU session F: ray generation with a gradient to the pixel coordinates -- unit test, the reference callers test, per-term diagnosis again; config 3 bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_gpu.py -m gpu -q -k "ray_gen" > gpurun_out/r04f_raygen.log 2>&1; echo "ray_gen rc=$?"; tail -2 gpurun_out/r04f_raygen.log
timeout 900 python -m pytest tests/test_reference_callers_gpu.py -m gpu -q > gpurun_out/r04f_reference_callers.log 2>&1; echo "reference callers rc=$?"; tail -4 gpurun_out/r04f_reference_callers.log
for a in "replica_sparf fp32" "llff_sparf bf16x3"; do echo "== $a"; timeout 300 python tests/tools/debug_callers.py $a 2>&1 | grep -v Warning | tail -22; done > gpurun_out/r04f_debug_callers.log 2>&1
grep -A4 "depth_cons\|^all\|^==" gpurun_out/r04f_debug_callers.log | cut -c1-200
timeout 600 python bench.py --config 3 --steps 20 --warmup 5 --no-psnr --no-other-sizes --no-other-modes --no-cpu-baseline --no-live-parity --no-roofline > gpurun_out/r04f_bench_c3.json 2> gpurun_out/r04f_bench_c3.err
python -c "
import json; d = json.loads(open('gpurun_out/r04f_bench_c3.json').read().strip().splitlines()[-1]); print('config 3', d['value'], d['ms_per_step'])"
