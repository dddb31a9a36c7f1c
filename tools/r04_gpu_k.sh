#!/bin/bash
# round 4, GPU session K: wgrad rows-per-split A/B (partial-sum traffic), callers test + yardstick with the GT-poses settings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for r in 4096 6144 8192 12288 3072; do
  SPARF_WG_ROWS=$r timeout 300 python bench.py --steps 40 --warmup 5 --min-seconds 2 --no-psnr --no-other-sizes --no-other-modes --no-cpu-baseline --no-live-parity 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['roofline']['all_kernels']; print('rows/split $r', round(d['value']), round(d['ms_per_step'], 3), round(d['sustained']['ms_per_step_p50'], 3), 'wgrad', k['wgrad']['launch_ms'], d['final_loss'])"
done | tee gpurun_out/r04k_wgrad_rows_per_split.log
timeout 900 python -m pytest tests/test_reference_callers_gpu.py -m gpu -q > gpurun_out/r04k_reference_callers.log 2>&1; echo "callers rc=$?"; tail -3 gpurun_out/r04k_reference_callers.log
timeout 900 python tests/tools/reference_callers_yardstick.py --settings dtu_nerf > gpurun_out/r04k_yardstick.log 2>&1; grep -E "^dtu" gpurun_out/r04k_yardstick.log | cut -c1-400
python - <<'PY'
import json
y = json.load(open("gpurun_out/r04_reference_callers_yardstick.json"))
print({k: v for k, v in y["dtu_nerf"]["per_call"][0].items()})
PY
