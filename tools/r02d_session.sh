#!/bin/bash
# round-2 GPU session d: config-3 seed sweep (heavy-tailed inverse-depth gradients), config-2 step profile, nt-default numbers
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
echo "== config 3 seeds"; timeout 600 python tools/scale_parity.py --configs 3 --precisions fp32,bf16x3 --yardstick --seeds 0,1,2,3,4,5 --referee-device cuda:0 --out gpurun_out/r02d_parity_c3_seeds.json 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print(d['seed'], d['precision'], 'out %.1e gradL2worst %.1e all %.1e d_dir %.1e d_pose %.1e' % (d['outputs_worst'], d['grad_l2_worst'], d['grad_l2_all'], d['d_viewdirs'], d['d_pose']), 'REF32: ' + ('out %.1e grad %.1e' % (d['reference_fp32_vs_referee']['outputs_worst'], d['reference_fp32_vs_referee']['grad_l2_worst']) if 'reference_fp32_vs_referee' in d else '-'))
"
echo "== config 1 seeds (metric)"; timeout 600 python tools/scale_parity.py --configs 1,2 --precisions fp32,bf16x3 --yardstick --seeds 1,2 --referee-device cuda:0 --out gpurun_out/r02d_parity_c12_seeds.json 2>&1 | grep '^{' | cut -c1-330
echo "== kernel bench (nt default)"; for p in bf16 bf16x3; do timeout 200 python tools/kernel_bench.py $p 2>&1 | grep -E "^(fwd|dgrad |wgrad|pass)" | awk '{printf "%s %s %s | ", $1, $2, $3} END {print ""}'; done
echo "== config 2 profile"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02d_c2 -- python bench.py --config 2 --no-cpu-baseline --no-psnr --no-roofline --no-other-modes --steps 20 > gpurun_out/r02d_prof_c2.log 2>&1; tail -1 gpurun_out/r02d_prof_c2.log | cut -c1-200
python tools/prof_summary.py gpurun_out/prof/r02d_c2_results.db gpurun_out/r02d_c2_kernel_stats.csv; head -45 gpurun_out/r02d_c2_kernel_stats.csv | cut -c1-150,161-
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02d_c1 -- python bench.py --config 1 --no-cpu-baseline --no-psnr --no-roofline --no-other-modes --steps 20 > gpurun_out/r02d_prof_c1.log 2>&1; tail -1 gpurun_out/r02d_prof_c1.log | cut -c1-200
python tools/prof_summary.py gpurun_out/prof/r02d_c1_results.db gpurun_out/r02d_c1_kernel_stats.csv; head -12 gpurun_out/r02d_c1_kernel_stats.csv | cut -c1-150,161-
