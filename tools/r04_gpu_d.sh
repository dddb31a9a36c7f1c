#!/bin/bash
# round 4, GPU session D: rocprofv3 kernel trace of config 3 (far rows), the reference-vs-itself yardstick of the callers test, LLFF-shaped training test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r04d_c3 -- python bench.py --config 3 --steps 10 --warmup 3 --min-seconds 0 --no-psnr --no-other-sizes --no-other-modes --no-cpu-baseline --no-live-parity --no-roofline > gpurun_out/r04d_prof_c3.log 2>&1
DB=$(find gpurun_out/prof -name "r04d_c3*results.db" | head -1); echo "db: $DB"
python tools/prof_summary.py "$DB" gpurun_out/r04d_config3_kernel_stats.csv && head -24 gpurun_out/r04d_config3_kernel_stats.csv | cut -c1-150
rm -rf gpurun_out/prof
timeout 900 python -m pytest tests/test_training_gpu.py -m gpu -q -s > gpurun_out/r04d_training.log 2>&1; echo "training test rc=$?"; grep -E "PSNR|passed|failed" gpurun_out/r04d_training.log
timeout 1500 python tests/tools/reference_callers_yardstick.py > gpurun_out/r04d_yardstick.log 2>&1; echo "yardstick rc=$?"; grep -E "^dtu|^llff|^replica|wrote" gpurun_out/r04d_yardstick.log
