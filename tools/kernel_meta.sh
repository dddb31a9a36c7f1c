#!/bin/bash
# per-kernel register / scratch / LDS figures of the built translation units (code-object notes): a kernel that starts to spill
# shows up here as private_segment_fixed_size > 0 before it shows up as a slowdown on the GPU
LLVM=/opt/rocm/lib/llvm/bin
B=$(dirname "$0")/../sparf_amd/csrc/build${1:-}
T=$(mktemp -d)
for o in "$B"/*.o; do
  n=$(basename "$o" .o)
  $LLVM/llvm-objcopy --dump-section .hip_fatbin=$T/$n.fat "$o" 2>/dev/null || continue
  $LLVM/clang-offload-bundler --type=o --unbundle --input=$T/$n.fat --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/$n.co 2>/dev/null || continue
  $LLVM/llvm-readelf --notes $T/$n.co 2>/dev/null | python3 -c "
import sys, re
txt = sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s*(\S+)', blk) or [None, '?'])[1]
    name = g('name')
    print(f'$n: {name[:70]:70s} vgpr {g(\"vgpr_count\"):>4s} agpr {blk.split()[1] if blk.split() else \"?\":>4s} sgpr {g(\"sgpr_count\"):>4s} scratch {g(\"private_segment_fixed_size\"):>5s} lds {g(\"group_segment_fixed_size\"):>7s} spill_v {g(\"vgpr_spill_count\")} spill_s {g(\"sgpr_spill_count\")}')
"
done
rm -rf $T
