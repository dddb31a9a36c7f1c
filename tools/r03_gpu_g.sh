#!/bin/bash
# Round-3 GPU session G: the GPU suite on the final tree, and the chip's clocks / power sampled once a second while the
# sustained bench loop runs (evidence for the DVFS statement of DESIGN 3.2).
set -u
TAG=r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== pytest -m gpu -x"; timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}g_pytest.log 2>&1; tail -4 gpurun_out/${TAG}g_pytest.log; grep -n "^E " gpurun_out/${TAG}g_pytest.log | head
echo "== idle clocks"; timeout 30 rocm-smi --showclocks --showpower --showuse 2>&1 | grep -i "sclk\|mclk\|power\|busy" | head -8
for MODE in bf16x3 bf16 fp32; do
  echo "== clocks under load: $MODE"
  timeout 300 python bench.py --precision $MODE --min-seconds 12 --no-live-parity --no-other-modes --no-other-sizes --no-psnr --no-cpu-baseline > gpurun_out/${TAG}g_bench_$MODE.json 2> gpurun_out/${TAG}g_bench_$MODE.err &
  BP=$!
  : > gpurun_out/${TAG}_clocks_$MODE.log
  while kill -0 $BP 2>/dev/null; do
    (date +%s.%N; timeout 10 rocm-smi --showclocks --showpower --showuse 2>&1 | grep -i "sclk\|mclk\|power\|busy") >> gpurun_out/${TAG}_clocks_$MODE.log
    sleep 0.7
  done
  wait $BP
  cut -c1-200 gpurun_out/${TAG}g_bench_$MODE.json
  grep -i "sclk" gpurun_out/${TAG}_clocks_$MODE.log | sort | uniq -c | sort -rn | head -6
  grep -i "power" gpurun_out/${TAG}_clocks_$MODE.log | tail -4
done
du -sh gpurun_out
