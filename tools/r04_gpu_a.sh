#!/bin/bash
# round 4, GPU session A: reference callers on the HIP Graph, inverse-depth routing study (emulation), baseline bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_reference_callers_gpu.py -m gpu -q -x --timeout 600 > gpurun_out/r04a_reference_callers.log 2>&1
echo "reference callers rc=$?" | tee -a gpurun_out/r04a_reference_callers.log
tail -5 gpurun_out/r04a_reference_callers.log
timeout 600 python tests/tools/inverse_routing_study.py > gpurun_out/r04a_inverse_routing_study.log 2>&1
echo "routing study rc=$?"; tail -8 gpurun_out/r04a_inverse_routing_study.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-psnr --no-other-sizes --no-other-modes > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err
echo "bench rc=$?"; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r04a_bench.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "ms_per_step")}, d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"], d["cpu_baseline"].get("port"))
except Exception as e:
    print("bench parse failed", e)
PY
