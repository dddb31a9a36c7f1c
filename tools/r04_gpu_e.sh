#!/bin/bash
# round 4, GPU session E: why do the HIP fp32 gradients of the replica SPARF iteration sit 1.5e-2 from the reference's? (per-term diagnosis);
# chunked dgrad || wgrad schedule: bit-identity + timing sweep
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for a in "replica_sparf fp32" "replica_sparf bf16x3" "llff_sparf fp32"; do echo "== $a"; timeout 300 python tests/tools/debug_callers.py $a 2>&1 | grep -v Warning | tail -22; done > gpurun_out/r04e_debug_callers.log 2>&1
cat gpurun_out/r04e_debug_callers.log | cut -c1-400
echo "== overlap: bit-identity (pass digest) and timing"
for ov in 0 2,64 4,64 4,96 4,48 8,64 4,32; do
  SPARF_OVERLAP=$ov timeout 300 python bench.py --steps 30 --warmup 5 --min-seconds 2 --no-psnr --no-other-sizes --no-other-modes --no-cpu-baseline --no-live-parity --no-roofline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap $ov', round(d['value']), round(d['ms_per_step'], 3), round(d['sustained']['ms_per_step_p50'], 3), d['final_loss'])"
done
