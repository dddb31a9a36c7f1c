#!/bin/bash
# Round-3 GPU session F: what the driver runs at round end (pytest -m gpu -x, smoke, the default bench line), on the final build,
# plus the parity profile stamped with the kernel source hash.
set -u
TAG=r03
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== pytest -m gpu -x"; timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}f_pytest.log 2>&1; tail -4 gpurun_out/${TAG}f_pytest.log; grep -n "^E " gpurun_out/${TAG}f_pytest.log | head
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== parity at the BASELINE shapes"; timeout 900 python tests/tools/scale_parity.py --yardstick --referee-device cuda:0 --out gpurun_out/${TAG}_parity_scale.json 2>&1 | grep '^{' | cut -c1-160
echo "== bench"; T0=$(date +%s); timeout 900 python bench.py > gpurun_out/${TAG}_bench_bf16x3.json 2> gpurun_out/${TAG}_bench_bf16x3.err; echo "bench wall seconds: $(( $(date +%s) - T0 ))"; cut -c1-400 gpurun_out/${TAG}_bench_bf16x3.json; tail -2 gpurun_out/${TAG}_bench_bf16x3.err
du -sh gpurun_out
