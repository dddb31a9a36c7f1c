#!/bin/bash
# round 4: bf16x3 data-gradient kernel with the weight HEADS only (one MFMA per product, -DSP_X3_DGRAD_PARTS=1, library _d1) against the
# default (heads + tails, two MFMAs): GPU suite, parity at the benchmark shapes, kernel times, step time, 2000-step PSNR
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SPARF_LIB=$PWD/sparf_amd/libsparf_hip_d1.so
echo "== pytest -m gpu (variant)"; timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04ab_pytest_d1.log 2>&1; tail -8 gpurun_out/r04ab_pytest_d1.log | cut -c1-200
echo "== parity (variant)"; timeout 900 python tests/tools/scale_parity.py --precisions bf16x3 --referee-device cuda:0 --out gpurun_out/r04ab_parity_d1.json 2>&1 | grep '^{' | cut -c1-330
unset SPARF_LIB
echo "== kernels"; AB_PRECS="bf16x3" bash tools/ab_kernels.sh d1 2>&1 | tee gpurun_out/r04ab_kernel_ab_d1.log
echo "== step"; for L in default d1; do
  if [ "$L" = default ]; then unset SPARF_LIB; else export SPARF_LIB=$PWD/sparf_amd/libsparf_hip_$L.so; fi
  echo "step $L: $(timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-psnr --no-roofline --no-other-modes --no-other-sizes --no-live-parity --no-strong-leg 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), "rays/s", round(d["ms_per_step"],3), "ms")')"
done | tee gpurun_out/r04ab_step_d1.log
export SPARF_LIB=$PWD/sparf_amd/libsparf_hip_d1.so
echo "== psnr (variant)"; timeout 900 python tests/tools/psnr_curve.py --config 1 --steps 2000 --modes bf16x3 --out gpurun_out/r04ab_psnr_d1.json --quiet 2>&1 | tail -2
python -c "
import json
d=json.load(open('gpurun_out/r04ab_psnr_d1.json'))
print('final', d['final'], 'delta', d['psnr_delta_vs_reference'])
print(json.dumps(d.get('photometric_grad_error_vs_float64_referee'))[:800])"
