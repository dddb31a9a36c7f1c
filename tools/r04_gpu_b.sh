#!/bin/bash
# round 4, GPU session B: far rows (inverse-depth routing) -- unit tests, config-3 parity, reference callers, whole GPU suite, config 1 / 3 bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_far_rows_gpu.py -m gpu -q -x -s > gpurun_out/r04b_far_rows.log 2>&1; echo "far rows rc=$?"; tail -4 gpurun_out/r04b_far_rows.log
timeout 900 python -m pytest tests/test_scale_gpu.py -m gpu -q -k "3-" -s > gpurun_out/r04b_scale_c3.log 2>&1; echo "scale c3 rc=$?"; tail -4 gpurun_out/r04b_scale_c3.log
timeout 900 python -m pytest tests/test_reference_callers_gpu.py -m gpu -q > gpurun_out/r04b_reference_callers.log 2>&1; echo "reference callers rc=$?"; tail -6 gpurun_out/r04b_reference_callers.log
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_reference_callers_gpu.py --deselect tests/test_far_rows_gpu.py > gpurun_out/r04b_gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -6 gpurun_out/r04b_gpu_suite.log
for c in 3 1; do
  timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-psnr --no-other-sizes --no-other-modes --no-cpu-baseline --no-live-parity > gpurun_out/r04b_bench_c$c.json 2> gpurun_out/r04b_bench_c$c.err
  echo "bench c$c rc=$?"; python -c "
import json; d = json.loads(open('gpurun_out/r04b_bench_c$c.json').read().strip().splitlines()[-1]); print('config $c', d['value'], d['ms_per_step'])"
done
SPARF_INVERSE_DEPTH_PRECISION=fp32 timeout 600 python bench.py --config 3 --steps 10 --warmup 3 --no-psnr --no-other-sizes --no-other-modes --no-cpu-baseline --no-live-parity --no-roofline > gpurun_out/r04b_bench_c3_wholefp32.json 2>/dev/null
SPARF_INVERSE_DEPTH_PRECISION=bf16x3 timeout 600 python bench.py --config 3 --steps 20 --warmup 5 --no-psnr --no-other-sizes --no-other-modes --no-cpu-baseline --no-live-parity --no-roofline > gpurun_out/r04b_bench_c3_nox.json 2>/dev/null
python - <<'PY'
import json
for f in ("r04b_bench_c3_wholefp32", "r04b_bench_c3_nox"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e:
        print(f, "failed", e)
PY
