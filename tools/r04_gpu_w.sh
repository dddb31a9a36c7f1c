#!/bin/bash
# round 4, 8-bit areas: exactness tests, kernel timings (phased conversion = default build, interleaved = _il build), step times of the four modes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_q8_saves_gpu.py -q -s > gpurun_out/r04w_q8_tests.log 2>&1; echo "q8 tests rc=$?"; grep -n "entries differ\|q8 vs plane\|gradient vs plane\|dequantised\|passed\|failed\|Error" gpurun_out/r04w_q8_tests.log | head -40
AB_PRECS="bf16 bf16+q8" bash tools/ab_kernels.sh il > gpurun_out/r04w_q8_kernel_ab.log 2>&1; cat gpurun_out/r04w_q8_kernel_ab.log
for prec in bf16 bf16+q8 bf16x3 bf16x3+q8; do
  timeout 600 python bench.py --precision $prec --steps 60 --warmup 10 --no-other-modes --no-strong-leg 2> gpurun_out/r04w_bench_$prec.err | tail -1 > gpurun_out/r04w_bench_$prec.json
  python -c "import json,sys; d=json.load(open('gpurun_out/r04w_bench_$prec.json')); print('$prec', round(d['value']), round(d['ms_per_step'],3), {k: v['launch_ms'] for k, v in d.get('roofline', {}).get('all_kernels', {}).items()})" || tail -5 gpurun_out/r04w_bench_$prec.err
done
