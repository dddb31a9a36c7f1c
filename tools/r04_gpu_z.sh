#!/bin/bash
# round 4: where the eager configs (3, 4) leave the GPU idle -- kernel timeline of the bench command, gaps by neighbouring kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/trace
for c in 3 4; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o c$c -- python bench.py --config $c --steps 20 --warmup 3 --min-seconds 0 --no-psnr --no-other-sizes --no-other-modes --no-cpu-baseline --no-live-parity --no-roofline --no-strong-leg > gpurun_out/trace/c$c.log 2>&1
  echo "== config $c"; python tools/gap_analysis.py $(ls gpurun_out/trace/*c${c}_kernel_trace.csv | head -1) 15
done > gpurun_out/r04z_gap_analysis.log 2>&1
rm -rf gpurun_out/trace
cat gpurun_out/r04z_gap_analysis.log
