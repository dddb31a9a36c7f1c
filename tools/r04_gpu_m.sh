#!/bin/bash
# round 4, GPU session M: far rows with transplanted saves (no far backward) -- tests, config 3 parity + bench + kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
timeout 600 python -m pytest tests/test_far_rows_gpu.py -m gpu -q -x -s > gpurun_out/r04m_far_rows.log 2>&1; echo "far rows rc=$?"; grep -E "rendered error|d params|d center|d dirs|passed|failed|Error" gpurun_out/r04m_far_rows.log | tail -12
timeout 900 python -m pytest tests/test_scale_gpu.py -m gpu -q -k "3-" -s > gpurun_out/r04m_scale_c3.log 2>&1; echo "scale c3 rc=$?"; tail -3 gpurun_out/r04m_scale_c3.log
timeout 900 python -m pytest tests/test_properties.py tests/test_training_gpu.py tests/test_reference_callers_gpu.py tests/test_callers_gpu.py tests/test_graph_capture_gpu.py -m gpu -q > gpurun_out/r04m_more.log 2>&1; echo "more rc=$?"; tail -3 gpurun_out/r04m_more.log
for E in "" "SPARF_FAR_SAMPLES=1" "SPARF_FAR_SAMPLES=4"; do for B in "" "--batched"; do
  echo "config 3 [$E] $B: $(env $E timeout 300 python bench.py --config 3 $B --steps 20 --warmup 5 --min-seconds 0 --no-cpu-baseline --no-psnr --no-roofline --no-other-modes --no-other-sizes --no-live-parity 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), "rays/s", round(d["ms_per_step"],2), "ms")')"
done; done | tee gpurun_out/r04m_config3_variants.log
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r04m_c3 -- python bench.py --config 3 --steps 20 --warmup 3 --min-seconds 0 --no-psnr --no-other-sizes --no-other-modes --no-cpu-baseline --no-live-parity --no-roofline > gpurun_out/r04m_prof_c3.log 2>&1
python tools/prof_summary.py gpurun_out/prof/r04m_c3_results.db gpurun_out/r04m_config3_kernel_stats.csv; head -14 gpurun_out/r04m_config3_kernel_stats.csv | cut -c1-110,160-
rm -rf gpurun_out/prof
