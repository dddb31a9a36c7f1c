#!/bin/bash
# One GPU session producing everything the round commits under profiles/:
#   bash tools/round_gpu.sh <tag>      (on the GPU box, from the repo root)
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== bench bf16 (default)"; timeout 600 python bench.py > gpurun_out/${TAG}_bench_bf16.json 2> gpurun_out/${TAG}_bench_bf16.err; tail -c 1500 gpurun_out/${TAG}_bench_bf16.json
echo "== bench fp32"; timeout 600 python bench.py --precision fp32 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_fp32.json 2> gpurun_out/${TAG}_bench_fp32.err; cut -c1-400 gpurun_out/${TAG}_bench_fp32.json
echo "== rocprofv3 kernel trace of the bench command"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG}_bf16 -- python bench.py --no-cpu-baseline > gpurun_out/${TAG}_prof_bench.log 2>&1
python tools/prof_summary.py gpurun_out/prof/${TAG}_bf16_results.db gpurun_out/${TAG}_bf16_kernel_stats.csv
echo "== PMC passes"
bash tools/pmc_profile.sh $TAG bf16 | tail -4
python tools/pmc_summary.py gpurun_out/pmc_$TAG gpurun_out/${TAG}_pmc_bf16 | cut -c1-250
