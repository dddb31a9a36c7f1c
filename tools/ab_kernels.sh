#!/bin/bash
# Same-box A/B of kernel variants (box-to-box variance is ~10 %): per-kernel timings of every
# sparf_amd/libsparf_hip_<tag>.so given on the command line next to the default build.
#   bash tools/ab_kernels.sh nt prio          (on the GPU box, from the repo root)
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
for tag in default "$@"; do
  if [ "$tag" = default ]; then unset SPARF_LIB; else export SPARF_LIB=$PWD/sparf_amd/libsparf_hip_$tag.so; fi
  for prec in ${AB_PRECS:-bf16 bf16x3}; do
    echo "== rep $rep lib $tag prec $prec"
    timeout 300 python tools/kernel_bench.py $prec 2>&1 | grep -E "^(fwd|dgrad |wgrad|pass)" | awk '{printf "%s %s %s | ", $1, $2, $3} END {print ""}'
  done
done
done
