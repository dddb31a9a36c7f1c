#!/bin/bash
# Round-3 GPU session I: session C again on the final kernels (staged tile inputs, slot balance) without the PSNR curves (the
# arithmetic is bit-identical: tools/pass_digest.py against the previous build), i.e. everything under profiles/r03_* but those.
#   bash tools/r03_gpu_i.sh        (on the GPU box, from the repo root)
set -u
TAG=r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
# (sparf_amd/libsparf_hip_base.so = `python tools/build_variant.py <revision> base`; without it only the current digests are logged)
echo "== bit-identity with the previous kernels"; for P in bf16x3 bf16 fp32; do for LIB in "" $(ls $PWD/sparf_amd/libsparf_hip_base.so 2>/dev/null); do SPARF_ABI_ANY=1 SPARF_LIB=$LIB timeout 200 python tools/pass_digest.py $P 2>&1 | tail -1; done; done | tee gpurun_out/${TAG}_pass_digest.log
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; tail -5 gpurun_out/${TAG}_pytest.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== parity at the BASELINE shapes"; timeout 900 python tests/tools/scale_parity.py --yardstick --referee-device cuda:0 --out gpurun_out/${TAG}_parity_scale.json 2>&1 | grep '^{' | cut -c1-220
echo "== config 3, six seeds"; timeout 600 python tests/tools/scale_parity.py --configs 3 --precisions 'bf16x3,bf16x3!,fp32' --seeds 0,1,2,3,4,5 --referee-device cuda:0 --out gpurun_out/${TAG}_parity_config3_six_seeds.json 2>&1 | grep '^{' | cut -c1-150
echo "== next-2: batched vs separate"
for P in bf16x3 bf16; do timeout 200 python tools/batch_bench.py $P 2>&1 | grep "rays/s"; done | tee gpurun_out/${TAG}_batch_bench.log
for E in "SPARF_INVERSE_DEPTH_PRECISION=bf16x3" ""; do for R in 4096 2048 1024; do for B in "" "--batched"; do
  echo "config 3 [$E] rays $R $B: $(env $E timeout 300 python bench.py --config 3 --rays $R $B --steps 15 --warmup 3 --min-seconds 0 --no-cpu-baseline --no-psnr --no-roofline --no-other-modes --no-other-sizes --no-live-parity 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), "rays/s", round(d["ms_per_step"],2), "ms")')"
done; done; done | tee -a gpurun_out/${TAG}_batch_bench.log
echo "== bench (default = bf16x3 headline)"; timeout 900 python bench.py > gpurun_out/${TAG}_bench_bf16x3.json 2> gpurun_out/${TAG}_bench_bf16x3.err; cut -c1-500 gpurun_out/${TAG}_bench_bf16x3.json; tail -2 gpurun_out/${TAG}_bench_bf16x3.err
echo "== bench bf16"; timeout 600 python bench.py --precision bf16 --no-cpu-baseline --no-other-modes --no-psnr > gpurun_out/${TAG}_bench_bf16.json 2> gpurun_out/${TAG}_bench_bf16.err; cut -c1-300 gpurun_out/${TAG}_bench_bf16.json
for c in 2 3 4; do
  echo "== bench config $c"; timeout 400 python bench.py --config $c --no-cpu-baseline --no-psnr --no-roofline --no-other-sizes --no-live-parity --steps 20 > gpurun_out/${TAG}_bench_c$c.json 2> gpurun_out/${TAG}_bench_c$c.err; cut -c1-330 gpurun_out/${TAG}_bench_c$c.json
done
echo "== batch-size sweep (bf16)"; for R in 512 1024 2048 4096 8192 16384; do echo "bf16 rays $R: $(timeout 200 python bench.py --precision bf16 --rays $R --steps 30 --warmup 5 --min-seconds 0 --no-cpu-baseline --no-psnr --no-roofline --no-other-modes --no-other-sizes --no-live-parity 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), "rays/s", round(d["ms_per_step"],3), "ms")')"; done | tee gpurun_out/${TAG}_batch_size_sweep.log
for R in 8192 16384; do echo "bf16x3 rays $R: $(timeout 200 python bench.py --rays $R --steps 20 --warmup 3 --min-seconds 0 --no-cpu-baseline --no-psnr --no-roofline --no-other-modes --no-other-sizes --no-live-parity 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), "rays/s", round(d["ms_per_step"],3), "ms")')"; done | tee -a gpurun_out/${TAG}_batch_size_sweep.log
echo "== eval bench"; for P in bf16 bf16x3 fp32; do timeout 300 python tools/eval_bench.py $P 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/${TAG}_eval_bench.log
echo "== rocprofv3 kernel trace of the bench command"
for P in bf16x3 bf16; do
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG}_$P -- python bench.py --precision $P --no-cpu-baseline --no-other-modes --no-psnr --no-other-sizes --no-live-parity > gpurun_out/${TAG}_prof_bench_$P.log 2>&1
  python tools/prof_summary.py gpurun_out/prof/${TAG}_${P}_results.db gpurun_out/${TAG}_${P}_kernel_stats.csv; head -6 gpurun_out/${TAG}_${P}_kernel_stats.csv | cut -c1-120,160-
done
echo "== rocprofv3 kernel trace, 512-ray steps (eager)"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG}_r512 -- python bench.py --rays 512 --steps 200 --warmup 10 --min-seconds 0 --no-cpu-baseline --no-other-modes --no-psnr --no-other-sizes --no-live-parity --no-roofline > gpurun_out/${TAG}_prof_r512.log 2>&1
python tools/prof_summary.py gpurun_out/prof/${TAG}_r512_results.db gpurun_out/${TAG}_r512_kernel_stats.csv; head -12 gpurun_out/${TAG}_r512_kernel_stats.csv | cut -c1-100,160-
python -c "
import csv
rows=list(csv.DictReader(open('gpurun_out/${TAG}_r512_kernel_stats.csv')))
print('512-ray run: total kernel ms', sum(float(r['total_ns']) for r in rows)/1e6, 'launches', sum(int(r['calls']) for r in rows))"
grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_prof_r512.log | head -1
rm -rf gpurun_out/prof
echo "== PMC passes"
for P in bf16x3 bf16; do
  bash tools/pmc_profile.sh ${TAG}_$P $P | grep "pass "
  python tools/pmc_summary.py gpurun_out/pmc_${TAG}_$P gpurun_out/${TAG}_pmc_$P | grep "mlp_\|wgrad_kernel" | cut -c1-260
  rm -rf gpurun_out/pmc_${TAG}_$P
done
echo "== PMC deep (SQ counters), bf16x3"; bash tools/pmc_deep.sh bf16x3 gpurun_out/pmc_deep_bf16x3 > gpurun_out/${TAG}_pmc_deep_bf16x3.txt 2>&1; grep -A 30 "mlp_fwd_kernel<2, true>\|mlp_fwd_kernel<2,true>" gpurun_out/${TAG}_pmc_deep_bf16x3.txt | head -40; rm -rf gpurun_out/pmc_deep_bf16x3


du -sh gpurun_out
