#!/bin/bash
# round-2 GPU session c: new tests, extreme-depth debug, kernel A/B, strict bf16x3 variants, new bench configs
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== new tests"; timeout 900 python -m pytest tests/test_callers_gpu.py tests/test_parallel_gpu.py tests/test_graph_gpu.py -m gpu -q 2>&1 | tail -25
echo "== debug extreme"; timeout 400 python tools/debug_extreme.py 2>&1 | tail -8
echo "== ab"; bash tools/ab_kernels.sh nt prio ntprio
echo "== strict bf16x3 variants"
for tag in x3dg x3full; do
  export SPARF_LIB=$PWD/sparf_amd/libsparf_hip_$tag.so
  timeout 400 python tools/scale_parity.py --configs 1,2 --precisions bf16x3 --referee-device cuda:0 --out gpurun_out/r02c_parity_$tag.json 2>&1 | grep '^{' | cut -c1-420
  timeout 200 python tools/kernel_bench.py bf16x3 2>&1 | grep -E "^(fwd|dgrad |wgrad|pass)" | awk '{printf "%s %s %s | ", $1, $2, $3} END {print ""}'
  timeout 200 python bench.py --no-cpu-baseline --no-psnr --no-other-modes --no-roofline --steps 20 2>/dev/null | cut -c1-260
done
unset SPARF_LIB
echo "== bench default"; timeout 600 python bench.py > gpurun_out/r02c_bench.json 2> gpurun_out/r02c_bench.err; tail -c 5000 gpurun_out/r02c_bench.json; tail -3 gpurun_out/r02c_bench.err
for c in 2 3 4; do
  echo "== bench config $c"; timeout 300 python bench.py --config $c --no-cpu-baseline --no-psnr --no-roofline --steps 15 > gpurun_out/r02c_bench_c$c.json 2> gpurun_out/r02c_bench_c$c.err; cut -c1-1500 gpurun_out/r02c_bench_c$c.json; tail -3 gpurun_out/r02c_bench_c$c.err
done
echo "== bench config 3 batched"; timeout 300 python bench.py --config 3 --batched --no-cpu-baseline --no-psnr --no-roofline --no-other-modes --steps 15 2> gpurun_out/r02c_bench_c3b.err | tee gpurun_out/r02c_bench_c3b.json | cut -c1-400; tail -3 gpurun_out/r02c_bench_c3b.err
