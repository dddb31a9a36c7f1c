#!/bin/bash
# round 4, 8-bit weight-gradient kernel, same box: schedule 0 (conversion phase in front of a tile's MFMAs, per-wave step rows) against
# schedule 1 (shared step rows one tile ahead; _ph: conversion as a phase behind the MFMAs, _il: interleaved with them); exactness tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_q8_saves_gpu.py -q > gpurun_out/r04x_q8_tests.log 2>&1; echo "q8 tests rc=$?"; tail -3 gpurun_out/r04x_q8_tests.log
for rep in 1 2 3; do
for tag in default ph il; do
  if [ "$tag" = default ]; then unset SPARF_LIB; else export SPARF_LIB=$PWD/sparf_amd/libsparf_hip_$tag.so; fi
  echo "== rep $rep lib $tag $(timeout 300 python tools/kernel_bench.py bf16+q8 2>&1 | grep -E '^(wgrad|pass bwd)' | tr '\n' ' ')"
done
done > gpurun_out/r04x_wgrad_q8_schedules.log 2>&1
cat gpurun_out/r04x_wgrad_q8_schedules.log
