#!/bin/bash
# Round-3 GPU session B: re-validation after the dgrad hazard fix, the inverse-depth fallback, variant A/Bs (dot2 tail split,
# exact xyz columns off), rocprof comparison of batched vs separate render calls, the final PSNR curves.
set -u
TAG=r03b
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== render_batch tests, full output"; timeout 600 python -m pytest tests/test_callers_gpu.py tests/test_graph_gpu.py -k "render_batch or short_training" -q --tb=long > gpurun_out/${TAG}_pytest_batch.log 2>&1; tail -3 gpurun_out/${TAG}_pytest_batch.log
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; tail -6 gpurun_out/${TAG}_pytest.log
OK=1; grep -q " failed\| error" gpurun_out/${TAG}_pytest.log && OK=0
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== dot2 probe"; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/dot2_probe.hip -o /tmp/dot2_probe 2>/dev/null && /tmp/dot2_probe | tee tools/probes/dot2_probe.out; cp tools/probes/dot2_probe.out gpurun_out/${TAG}_dot2_probe.out
echo "== parity at the BASELINE shapes"; timeout 900 python tests/tools/scale_parity.py --yardstick --referee-device cuda:0 --out gpurun_out/${TAG}_parity_scale.json 2>&1 | grep '^{' | cut -c1-250
echo "== config 3, six seeds: bf16x3 (fp32 fallback), bf16x3! (opt-out), fp32"; timeout 600 python tests/tools/scale_parity.py --configs 3 --precisions 'bf16x3,bf16x3!,fp32' --seeds 0,1,2,3,4,5 --referee-device cuda:0 --out gpurun_out/${TAG}_parity_c3_seeds.json 2>&1 | grep '^{' | cut -c1-200
echo "== dot2 variant parity (configs 1, 3 opt-out)"; SPARF_ABI_ANY=1 SPARF_LIB=$PWD/sparf_amd/libsparf_hip_dot2.so timeout 300 python tests/tools/scale_parity.py --configs 1,3 --precisions 'bf16x3,bf16x3!' --referee-device cuda:0 --out gpurun_out/${TAG}_parity_dot2.json 2>&1 | grep '^{' | cut -c1-200
echo "== kernel A/B: default, dot2, noxyz, r02"; AB_PRECS=bf16x3 SPARF_ABI_ANY=1 bash tools/ab_kernels.sh dot2 noxyz r02 2>&1 | tee gpurun_out/${TAG}_ab.log
echo "== kernel A/B bf16: default vs r02"; AB_PRECS=bf16 SPARF_ABI_ANY=1 bash tools/ab_kernels.sh r02 2>&1 | tee -a gpurun_out/${TAG}_ab.log
echo "== rocprof: config 3 separate vs batched"
for B in "" "--batched"; do
  T=sep; [ -n "$B" ] && T=bat
  timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG}_c3_$T -- python bench.py --config 3 $B --steps 20 --warmup 3 --min-seconds 0 --no-cpu-baseline --no-psnr --no-roofline --no-other-modes --no-other-sizes --no-live-parity > gpurun_out/${TAG}_c3_$T.json 2> gpurun_out/${TAG}_c3_$T.err
  python tools/prof_summary.py gpurun_out/prof/${TAG}_c3_${T}_results.db gpurun_out/${TAG}_c3_${T}_kernel_stats.csv; head -14 gpurun_out/${TAG}_c3_${T}_kernel_stats.csv | cut -c1-110
  python -c "
import csv,sys
rows=list(csv.DictReader(open('gpurun_out/${TAG}_c3_${T}_kernel_stats.csv')))
print('$T total kernel ms', sum(float(r['total_ns']) for r in rows)/1e6, 'launches', sum(int(r['calls']) for r in rows))"
  cut -c1-200 gpurun_out/${TAG}_c3_$T.json
done
rm -rf gpurun_out/prof
if [ $OK = 1 ]; then
echo "== psnr curve, config 1"; timeout 700 python tests/tools/psnr_curve.py --config 1 --steps 2000 --max-seconds 560 --out gpurun_out/${TAG}_psnr_curve_c1.json 2>&1 | tail -3 | cut -c1-600
echo "== psnr curve, config 2"; timeout 700 python tests/tools/psnr_curve.py --config 2 --steps 2000 --max-seconds 560 --out gpurun_out/${TAG}_psnr_curve_c2.json 2>&1 | tail -3 | cut -c1-600
fi
echo "== bench config 3 (fallback / opt-out)"; for E in "" "SPARF_INVERSE_DEPTH_PRECISION=bf16x3"; do env $E timeout 300 python bench.py --config 3 --steps 15 --warmup 3 --min-seconds 0 --no-cpu-baseline --no-psnr --no-roofline --no-other-modes --no-other-sizes --no-live-parity 2>/dev/null | cut -c1-260; done | tee gpurun_out/${TAG}_bench_c3.log
echo "== bench"; timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cut -c1-400 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
du -sh gpurun_out
