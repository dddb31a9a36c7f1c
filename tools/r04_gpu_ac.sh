#!/bin/bash
# round 4: the eager configs with the 20-step default warm-up (contract region = steady state), and the warm-up sweep that motivated it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=r04
for W in 5 10 20 40; do
  echo "config 3, --warmup $W: $(python bench.py --config 3 --no-cpu-baseline --no-psnr --no-roofline --no-other-sizes --no-other-modes --no-live-parity --steps 20 --warmup $W --min-seconds 2 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), "rays/s in the 20 timed steps,", round(d["ms_per_step"],2), "ms; sustained", round(d["sustained"]["value"]))')"
done | tee gpurun_out/${TAG}_config3_warmup.log
for c in 3 4; do
  echo "== bench config $c"; timeout 400 python bench.py --config $c --no-cpu-baseline --no-psnr --no-roofline --no-other-sizes --no-live-parity --steps 20 > gpurun_out/${TAG}_bench_c$c.json 2> gpurun_out/${TAG}_bench_c$c.err; cut -c1-200 gpurun_out/${TAG}_bench_c$c.json
done
echo "== config 3: inverse-depth variants, separate / batched"
for E in "" "SPARF_INVERSE_DEPTH_PRECISION=fp32" "SPARF_INVERSE_DEPTH_PRECISION=bf16x3" "SPARF_FAR_SAMPLES=1" "SPARF_FAR_SAMPLES=4"; do for B in "" "--batched"; do
  echo "config 3 [$E] $B: $(env $E timeout 300 python bench.py --config 3 $B --steps 15 --min-seconds 0 --no-cpu-baseline --no-psnr --no-roofline --no-other-modes --no-other-sizes --no-live-parity 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), "rays/s", round(d["ms_per_step"],2), "ms")')"
done; done | tee gpurun_out/${TAG}_config3_variants.log
