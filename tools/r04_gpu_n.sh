#!/bin/bash
# round 4, GPU session N: far tiles by value (render_to_max under inverse depth) -- tests, config 3 / 1 bench, full GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_far_rows_gpu.py -m gpu -q -x > gpurun_out/r04n_far_rows.log 2>&1; echo "far rows rc=$?"; tail -3 gpurun_out/r04n_far_rows.log
timeout 900 python -m pytest tests/test_scale_gpu.py -m gpu -q -k "3-" -s > gpurun_out/r04n_scale_c3.log 2>&1; echo "scale c3 rc=$?"; grep -o '"to_max": {[^}]*}' gpurun_out/r04n_scale_c3.log | head -4; tail -2 gpurun_out/r04n_scale_c3.log
for c in 3 1; do
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-psnr --no-other-sizes --no-other-modes --no-cpu-baseline --no-live-parity --no-roofline 2>/dev/null | python -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c', round(d['value']), round(d['ms_per_step'],3), round(d['sustained']['ms_per_step_p50'],3))"
done
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_far_rows_gpu.py > gpurun_out/r04n_gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -5 gpurun_out/r04n_gpu_suite.log
