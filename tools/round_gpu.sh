#!/bin/bash
# One GPU session producing everything the round commits under profiles/:
#   bash tools/round_gpu.sh <tag>      (on the GPU box, from the repo root)
# Outputs (small files only; the rocprof databases are summarised and deleted, gpurun_out/ is capped at 64 MiB):
#   gpurun_out/<tag>_bench_bf16x3.json   the default bench line (headline mode + other modes + psnr + cpu baseline)
#   gpurun_out/<tag>_bench_c{2,3,4}.json BASELINE configs 2-4
#   gpurun_out/<tag>_{bf16x3,bf16}_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the bench command
#   gpurun_out/<tag>_pmc_{bf16x3,bf16}.{json,csv}     PMC passes (HBM bytes, MFMA busy, LDS conflicts)
#   gpurun_out/<tag>_parity_scale.json   parity at the BASELINE shapes (float64 referee)
set -u
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== parity at the BASELINE shapes"; timeout 900 python tests/tools/scale_parity.py --yardstick --referee-device cuda:0 --out gpurun_out/${TAG}_parity_scale.json 2>&1 | grep '^{' | cut -c1-260
echo "== bench (default = bf16x3 headline, other modes inside)"; timeout 900 python bench.py > gpurun_out/${TAG}_bench_bf16x3.json 2> gpurun_out/${TAG}_bench_bf16x3.err; tail -c 1500 gpurun_out/${TAG}_bench_bf16x3.json
echo "== bench bf16 (throughput mode)"; timeout 600 python bench.py --precision bf16 --no-cpu-baseline --no-other-modes --no-psnr > gpurun_out/${TAG}_bench_bf16.json 2> gpurun_out/${TAG}_bench_bf16.err; cut -c1-300 gpurun_out/${TAG}_bench_bf16.json
for c in 2 3 4; do
  echo "== bench config $c"; timeout 400 python bench.py --config $c --no-cpu-baseline --no-psnr --no-roofline --steps 20 > gpurun_out/${TAG}_bench_c$c.json 2> gpurun_out/${TAG}_bench_c$c.err; cut -c1-330 gpurun_out/${TAG}_bench_c$c.json
done
echo "== rocprofv3 kernel trace of the bench command"
for P in bf16x3 bf16; do
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG}_$P -- python bench.py --precision $P --no-cpu-baseline --no-other-modes --no-psnr > gpurun_out/${TAG}_prof_bench_$P.log 2>&1
  python tools/prof_summary.py gpurun_out/prof/${TAG}_${P}_results.db gpurun_out/${TAG}_${P}_kernel_stats.csv; head -6 gpurun_out/${TAG}_${P}_kernel_stats.csv | cut -c1-120,160-
done
rm -rf gpurun_out/prof
echo "== PMC passes"
for P in bf16x3 bf16; do
  bash tools/pmc_profile.sh ${TAG}_$P $P | grep "pass "
  python tools/pmc_summary.py gpurun_out/pmc_${TAG}_$P gpurun_out/${TAG}_pmc_$P | grep "mlp_\|wgrad_kernel" | cut -c1-260
  rm -rf gpurun_out/pmc_${TAG}_$P
done
du -sh gpurun_out
