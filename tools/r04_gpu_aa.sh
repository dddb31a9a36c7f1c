#!/bin/bash
# round 4: HBM traffic (PMC FETCH_SIZE / WRITE_SIZE) of the kernels under 8-bit save / gradient areas, and the full bench lines of the two modes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for P in bf16x3+q8 bf16+q8; do
  T=$(echo $P | tr '+' '_')
  bash tools/pmc_profile.sh r04_$T $P | grep "pass "
  python tools/pmc_summary.py gpurun_out/pmc_r04_$T gpurun_out/r04_pmc_$T | grep "mlp_\|wgrad_kernel" | cut -c1-300
  rm -rf gpurun_out/pmc_r04_$T
  timeout 600 python bench.py --precision $P --no-cpu-baseline --no-other-modes --no-psnr --no-other-sizes > gpurun_out/r04_bench_$T.json 2> gpurun_out/r04_bench_$T.err; cut -c1-200 gpurun_out/r04_bench_$T.json
done
