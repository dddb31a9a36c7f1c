#!/bin/bash
# round 4: what the weight-gradient kernels' MFMA phase costs -- the LDS fragment reads or the MFMAs (probe builds, wrong results)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for tag in default nofrag nomfma; do
  if [ "$tag" = default ]; then unset SPARF_LIB; else export SPARF_LIB=$PWD/sparf_amd/libsparf_hip_$tag.so; fi
  for prec in bf16 bf16+q8; do
    echo "== lib $tag prec $prec $(timeout 300 python tools/kernel_bench.py $prec 2>&1 | grep -E '^wgrad')"
  done
done > gpurun_out/r04r_wgrad_phase_probes.log 2>&1
cat gpurun_out/r04r_wgrad_phase_probes.log
