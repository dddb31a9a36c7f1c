#!/bin/bash
# round 4, GPU session C: far rows after the view-encoding layout fix -- tests, config 1 / 3 bench, rocprofv3 kernel trace of a config-3 run
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_far_rows_gpu.py -m gpu -q -x -s > gpurun_out/r04c_far_rows.log 2>&1; echo "far rows rc=$?"; grep -E "rendered error|d params|d center|d dirs|passed|failed" gpurun_out/r04c_far_rows.log | tail -12
timeout 900 python -m pytest tests/test_scale_gpu.py -m gpu -q -k "3-" -s > gpurun_out/r04c_scale_c3.log 2>&1; echo "scale c3 rc=$?"; tail -3 gpurun_out/r04c_scale_c3.log
timeout 900 python -m pytest tests/test_reference_callers_gpu.py -m gpu -q > gpurun_out/r04c_reference_callers.log 2>&1; echo "reference callers rc=$?"; tail -4 gpurun_out/r04c_reference_callers.log
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_reference_callers_gpu.py --deselect tests/test_far_rows_gpu.py > gpurun_out/r04c_gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -8 gpurun_out/r04c_gpu_suite.log
for c in 1 3 1; do
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-psnr --no-other-sizes --no-other-modes --no-cpu-baseline --no-live-parity --no-roofline > gpurun_out/r04c_bench_c$c.json 2> gpurun_out/r04c_bench_c$c.err
  python -c "
import json; d = json.loads(open('gpurun_out/r04c_bench_c$c.json').read().strip().splitlines()[-1]); print('config $c', d['value'], d['ms_per_step'], d.get('sustained', {}).get('ms_per_step_p50'))"
done
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04c_prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --config 3 --steps 10 --warmup 3 --min-seconds 0 --no-psnr --no-other-sizes --no-other-modes --no-cpu-baseline --no-live-parity --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/r04c_prof_c3.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r04c_prof_c3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04c_config3_kernel_stats.csv; head -25 gpurun_out/r04c_config3_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/r04c_prof_c3
