"""Issue-slot histogram of a fused-MLP kernel: how many non-MFMA instructions sit between consecutive MFMAs.

    python tools/mfma_gap_hist.py sparf_amd/csrc/mlp_fwd_x3_train.hip [-DFLAG ...]

Compiles the translation unit to gfx950 assembly (device only) and walks the instruction stream in program order.
A wave alone on its SIMD hides ~5 issue slots behind one v_mfma_f32_32x32x16_bf16 (32 cycles; MI355X_MICROARCH.md):
a gap of g instructions costs about max(32, c * (g + 1)) cycles with c = 4.5-5, so the mean gap says little and the
histogram says where the matrix pipe starves.  (Straight-line estimate: loops, e.g. the 15-iteration encoding
loop, count once.)"""
import collections
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gaps_of(asm):
    gaps, cur, started = [], 0, False
    for line in asm.split("\n"):
        t = line.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        if t.split()[0].startswith("v_mfma"):
            if started:
                gaps.append(cur)
            started, cur = True, 0
        elif started:
            cur += 1
    return gaps


def main():
    src, flags = sys.argv[1], sys.argv[2:]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-fconstexpr-steps=200000000",
                               "-I" + os.path.join(ROOT, "sparf_amd", "csrc"), "--cuda-device-only", "-S", "-o", out, src] + flags,
                              stderr=subprocess.DEVNULL)
        gaps = gaps_of(open(out).read())
    n = len(gaps)
    h = collections.Counter(gaps)
    print(f"{n + 1} MFMAs, {sum(gaps)} other instructions, mean gap {sum(gaps) / n:.2f}")
    edges = [(0, 0), (1, 2), (3, 5), (6, 8), (9, 15), (16, 63), (64, 10 ** 9)]
    for lo, hi in edges:
        c = sum(v for g, v in h.items() if lo <= g <= hi)
        stall = sum(max(0.0, 5.0 * (g + 1) - 32.0) * v for g, v in h.items() if lo <= g <= hi)
        print(f"  gap {lo:3d}..{hi if hi < 10 ** 9 else 'inf':>3}: {c:5d} gaps   modelled stall {stall / 1e3:6.1f} k cycles")
    for c in (4.0, 4.5, 5.0):
        print(f"  model c = {c}: {sum(max(32.0, c * (g + 1)) for g in gaps) / n:.1f} cycles per MFMA")


if __name__ == "__main__":
    main()
