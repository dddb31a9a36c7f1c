#!/bin/bash
# Round-3 GPU session D: new tests (hipGraph capture, segments from the C++ host), cross-layer deferred epilogue A/B.
set -u
TAG=r03d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; tail -6 gpurun_out/${TAG}_pytest.log; grep -n "^E " gpurun_out/${TAG}_pytest.log | head -20
echo "== XL variant parity (config 1 all, config 3 opt-out)"; SPARF_ABI_ANY=1 SPARF_LIB=$PWD/sparf_amd/libsparf_hip_xl.so timeout 300 python tests/tools/scale_parity.py --configs 1,3 --precisions 'bf16x3,bf16x3!' --referee-device cuda:0 --out gpurun_out/${TAG}_parity_xl.json 2>&1 | grep '^{' | cut -c1-260
echo "== XL variant: forward tests"; SPARF_ABI_ANY=1 SPARF_LIB=$PWD/sparf_amd/libsparf_hip_xl.so timeout 600 python -m pytest tests/test_hip_gpu.py tests/test_graph_gpu.py -m gpu -q 2>&1 | tail -4
echo "== kernel A/B: default vs xl"; AB_PRECS=bf16x3 SPARF_ABI_ANY=1 bash tools/ab_kernels.sh xl 2>&1 | tee gpurun_out/${TAG}_ab.log
for L in "" "$PWD/sparf_amd/libsparf_hip_xl.so"; do echo "bench lib [$L]: $(SPARF_ABI_ANY=1 SPARF_LIB=$L timeout 300 python bench.py --steps 50 --warmup 5 --min-seconds 2 --no-cpu-baseline --no-psnr --no-roofline --no-other-modes --no-other-sizes --no-live-parity 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]), "rays/s", round(d["ms_per_step"],3), "ms", round(d["sustained"]["value"]))')"; done | tee -a gpurun_out/${TAG}_ab.log
du -sh gpurun_out
