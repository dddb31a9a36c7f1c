#!/bin/bash
# Round-3 GPU session A: correctness of the refactored kernels (tile-block areas, FIFO masks, exact raw-coordinate
# columns), parity at the BASELINE shapes, same-box kernel A/B against the round-2 build, and the long side-by-side
# PSNR runs.   bash tools/r03_gpu_a.sh     (on the GPU box, from the repo root)
set -u
TAG=r03a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/${TAG}_pytest.log; tail -4 gpurun_out/${TAG}_pytest.log
OK=1; grep -q " failed\| error" gpurun_out/${TAG}_pytest.log && OK=0
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== parity at the BASELINE shapes"; timeout 900 python tests/tools/scale_parity.py --yardstick --referee-device cuda:0 --out gpurun_out/${TAG}_parity_scale.json 2>&1 | grep '^{' | cut -c1-330
echo "== config 3, six seeds, bf16x3 + fp32"; timeout 600 python tests/tools/scale_parity.py --configs 3 --precisions bf16x3,fp32 --seeds 0,1,2,3,4,5 --referee-device cuda:0 --out gpurun_out/${TAG}_parity_c3_seeds.json 2>&1 | grep '^{' | cut -c1-200
echo "== kernel A/B vs the round-2 build"; SPARF_ABI_ANY=1 bash tools/ab_kernels.sh r02 2>&1 | tee gpurun_out/${TAG}_ab.log
if [ $OK = 1 ]; then PS_ENV=""; else echo "!! GPU tests failed: PSNR runs use the round-2 kernels"; PS_ENV="SPARF_ABI_ANY=1 SPARF_LIB=$PWD/sparf_amd/libsparf_hip_r02.so"; fi
echo "== psnr curve, config 1"; env $PS_ENV timeout 700 python tests/tools/psnr_curve.py --config 1 --steps 2000 --max-seconds 560 --out gpurun_out/${TAG}_psnr_curve_c1.json 2>&1 | tail -14 | cut -c1-400
echo "== psnr curve, config 2"; env $PS_ENV timeout 700 python tests/tools/psnr_curve.py --config 2 --steps 2000 --max-seconds 560 --out gpurun_out/${TAG}_psnr_curve_c2.json 2>&1 | tail -14 | cut -c1-400
echo "== next-2: batched vs separate render calls"
for P in bf16x3 bf16; do timeout 200 python tools/batch_bench.py $P 2>&1 | grep "rays/s"; done | tee gpurun_out/${TAG}_batch_bench.log
for R in 4096 2048 1024; do for B in "" "--batched"; do
  echo "config 3 rays $R $B: $(timeout 300 python bench.py --config 3 --rays $R $B --steps 15 --warmup 3 --min-seconds 0 --no-cpu-baseline --no-psnr --no-roofline --no-other-modes --no-other-sizes --no-live-parity 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), "rays/s", round(d["ms_per_step"],2), "ms")')"
done; done | tee gpurun_out/${TAG}_batched_c3.log
echo "== bench"; timeout 600 python bench.py --no-psnr > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cut -c1-600 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
du -sh gpurun_out
