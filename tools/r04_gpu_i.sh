#!/bin/bash
# Round-4 GPU session I: everything under profiles/r04_* that the documents quote, from the shipped kernels, in one call.
#   bash tools/r04_gpu_i.sh        (on the GPU box, from the repo root)
set -u
TAG=r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
echo "== pytest -m gpu"; timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; tail -4 gpurun_out/${TAG}_pytest.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v Warn | tail -3
echo "== parity at the BASELINE shapes"; timeout 900 python tests/tools/scale_parity.py --yardstick --referee-device cuda:0 --out gpurun_out/${TAG}_parity_scale.json 2>&1 | grep '^{' | cut -c1-220
echo "== config 3, six seeds"; timeout 900 python tests/tools/scale_parity.py --configs 3 --precisions 'bf16x3,bf16x3#,bf16x3!,fp32' --seeds 0,1,2,3,4,5 --referee-device cuda:0 --out gpurun_out/${TAG}_parity_config3_six_seeds.json 2>&1 | grep '^{' | cut -c1-170
echo "== bench (default = bf16x3 headline)"; timeout 900 python bench.py > gpurun_out/${TAG}_bench_bf16x3.json 2> gpurun_out/${TAG}_bench_bf16x3.err; cut -c1-400 gpurun_out/${TAG}_bench_bf16x3.json; tail -2 gpurun_out/${TAG}_bench_bf16x3.err
echo "== 8-bit save modes: parity at config 1, kernels, step time"
timeout 600 python tests/tools/scale_parity.py --configs 1 --precisions 'bf16x3,bf16x3+q8,bf16,bf16+q8' --referee-device cuda:0 --out gpurun_out/${TAG}_parity_q8.json 2>&1 | grep '^{' | cut -c1-260
AB_PRECS="bf16 bf16+q8 bf16x3 bf16x3+q8" bash tools/ab_kernels.sh > gpurun_out/${TAG}_q8_kernel_ab.log 2>&1; cat gpurun_out/${TAG}_q8_kernel_ab.log
for P in bf16 bf16+q8 bf16x3 bf16x3+q8; do
  echo "step $P: $(timeout 300 python bench.py --precision $P --steps 60 --warmup 10 --no-cpu-baseline --no-psnr --no-roofline --no-other-modes --no-other-sizes --no-live-parity --no-strong-leg 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), "rays/s", round(d["ms_per_step"],3), "ms")')"
done | tee gpurun_out/${TAG}_q8_step_times.log
echo "== bench bf16"; timeout 600 python bench.py --precision bf16 --no-cpu-baseline --no-other-modes --no-psnr > gpurun_out/${TAG}_bench_bf16.json 2> gpurun_out/${TAG}_bench_bf16.err; cut -c1-200 gpurun_out/${TAG}_bench_bf16.json
for c in 2 3 4; do
  echo "== bench config $c"; timeout 400 python bench.py --config $c --no-cpu-baseline --no-psnr --no-roofline --no-other-sizes --no-live-parity --steps 20 > gpurun_out/${TAG}_bench_c$c.json 2> gpurun_out/${TAG}_bench_c$c.err; cut -c1-200 gpurun_out/${TAG}_bench_c$c.json
done
echo "== config 3: inverse-depth variants, separate / batched"
for E in "" "SPARF_INVERSE_DEPTH_PRECISION=fp32" "SPARF_INVERSE_DEPTH_PRECISION=bf16x3" "SPARF_FAR_SAMPLES=1" "SPARF_FAR_SAMPLES=4"; do for B in "" "--batched"; do
  echo "config 3 [$E] $B: $(env $E timeout 300 python bench.py --config 3 $B --steps 15 --warmup 3 --min-seconds 0 --no-cpu-baseline --no-psnr --no-roofline --no-other-modes --no-other-sizes --no-live-parity 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), "rays/s", round(d["ms_per_step"],2), "ms")')"
done; done | tee gpurun_out/${TAG}_config3_variants.log
echo "== eval bench"; for P in bf16 bf16x3 fp32; do timeout 300 python tools/eval_bench.py $P 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/${TAG}_eval_bench.log
echo "== rocprofv3 kernel trace of the bench command"
for P in bf16x3; do
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG}_$P -- python bench.py --precision $P --no-cpu-baseline --no-other-modes --no-psnr --no-other-sizes --no-live-parity > gpurun_out/${TAG}_prof_bench_$P.log 2>&1
  python tools/prof_summary.py gpurun_out/prof/${TAG}_${P}_results.db gpurun_out/${TAG}_${P}_kernel_stats.csv; head -8 gpurun_out/${TAG}_${P}_kernel_stats.csv | cut -c1-120,160-
done
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG}_c3 -- python bench.py --config 3 --steps 20 --warmup 3 --min-seconds 0 --no-psnr --no-other-sizes --no-other-modes --no-cpu-baseline --no-live-parity --no-roofline > gpurun_out/${TAG}_prof_c3.log 2>&1
python tools/prof_summary.py gpurun_out/prof/${TAG}_c3_results.db gpurun_out/${TAG}_config3_kernel_stats.csv; head -14 gpurun_out/${TAG}_config3_kernel_stats.csv | cut -c1-120,160-
rm -rf gpurun_out/prof
echo "== PMC passes"
for P in bf16x3; do
  bash tools/pmc_profile.sh ${TAG}_$P $P | grep "pass "
  python tools/pmc_summary.py gpurun_out/pmc_${TAG}_$P gpurun_out/${TAG}_pmc_$P | grep "mlp_\|wgrad_kernel" | cut -c1-260
  rm -rf gpurun_out/pmc_${TAG}_$P
done
du -sh gpurun_out
