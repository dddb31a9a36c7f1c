#!/bin/bash
# One GPU session producing everything the round commits under profiles/:
#   bash tools/round_gpu.sh <tag>      (on the GPU box, from the repo root)
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== bench (default = bf16x3 headline, other modes inside)"; timeout 900 python bench.py > gpurun_out/${TAG}_bench_bf16x3.json 2> gpurun_out/${TAG}_bench_bf16x3.err; tail -c 2500 gpurun_out/${TAG}_bench_bf16x3.json
echo "== bench bf16 (throughput mode)"; timeout 600 python bench.py --precision bf16 --no-cpu-baseline --no-other-modes > gpurun_out/${TAG}_bench_bf16.json 2> gpurun_out/${TAG}_bench_bf16.err; cut -c1-300 gpurun_out/${TAG}_bench_bf16.json
echo "== rocprofv3 kernel trace of the bench command"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG}_bf16x3 -- python bench.py --no-cpu-baseline --no-other-modes > gpurun_out/${TAG}_prof_bench.log 2>&1
python tools/prof_summary.py gpurun_out/prof/${TAG}_bf16x3_results.db gpurun_out/${TAG}_bf16x3_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG}_bf16 -- python bench.py --precision bf16 --no-cpu-baseline --no-other-modes > gpurun_out/${TAG}_prof_bench_bf16.log 2>&1
python tools/prof_summary.py gpurun_out/prof/${TAG}_bf16_results.db gpurun_out/${TAG}_bf16_kernel_stats.csv
echo "== PMC passes"
for P in bf16x3 bf16; do
  bash tools/pmc_profile.sh ${TAG}_$P $P | grep "pass "
  python tools/pmc_summary.py gpurun_out/pmc_${TAG}_$P gpurun_out/${TAG}_pmc_$P | grep "mlp_\|wgrad_kernel" | cut -c1-260
done
