#!/bin/bash
# round 4, GPU session L: wgrad rows-per-split sweep, larger splits; same box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for r in 4096 12288 16384 24576 32768 49152 4096 16384; do
  SPARF_WG_ROWS=$r timeout 300 python bench.py --steps 40 --warmup 5 --min-seconds 2 --no-psnr --no-other-sizes --no-other-modes --no-cpu-baseline --no-live-parity 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['roofline']['all_kernels']; print('rows/split $r', round(d['value']), round(d['ms_per_step'], 3), round(d['sustained']['ms_per_step_p50'], 3), 'wgrad', k['wgrad']['launch_ms'])"
done | tee gpurun_out/r04l_wgrad_rows_per_split.log
for r in 4096 16384; do for c in 2 4; do
  SPARF_WG_ROWS=$r timeout 300 python bench.py --config $c --steps 20 --warmup 5 --min-seconds 0 --no-psnr --no-other-sizes --no-other-modes --no-cpu-baseline --no-live-parity --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c rows/split $r', round(d['value']), round(d['ms_per_step'], 3))"
done; done | tee -a gpurun_out/r04l_wgrad_rows_per_split.log
for r in 4096 16384; do SPARF_WG_ROWS=$r timeout 300 python bench.py --precision bf16 --steps 40 --warmup 5 --min-seconds 2 --no-psnr --no-other-sizes --no-other-modes --no-cpu-baseline --no-live-parity 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['roofline']['all_kernels']; print('bf16 rows/split $r', round(d['value']), round(d['ms_per_step'], 3), 'wgrad', k['wgrad']['launch_ms'])"; done | tee -a gpurun_out/r04l_wgrad_rows_per_split.log
