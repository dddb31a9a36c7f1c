"""Build libsparf_hip.so in-tree with hipcc (gfx950 only).

`python -m sparf_amd.build` or `sparf_amd.build.build()`.  Translation units are compiled
in parallel; objects land in sparf_amd/csrc/build/, the library next to this file (both
git-ignored, but shipped to the GPU box by gpurun)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libsparf_hip.so")
# (the slow units first: they are compiled in parallel and set the build's wall time)
SOURCES = ["mlp_fwd_fp32_train.hip", "mlp_fwd_fp32_infer.hip", "mlp_fwd_x3_train.hip", "mlp_fwd_x3_train_q8.hip", "mlp_fwd_x3_infer.hip", "mlp_bwd.hip", "mlp_bwd_fp32.hip", "mlp_bwd_x3.hip", "mlp_bwd_x3w4.hip", "mlp_bwd_q8.hip",
           "mlp_fwd_bf16_train.hip", "mlp_fwd_bf16_train_q8.hip", "mlp_fwd_bf16_infer.hip", "wgrad.hip", "api.hip", "mlp_fwd.hip", "ray_ops.hip", "pack.hip", "optim.hip", "calib.hip",
           "tables.cpp"]
HEADERS = ["layout.h", "streams.h", "mlp_dev.h", "mlp_fwd_impl.h", "mlp_bwd_impl.h", "kernels.h", os.path.join("..", "..", "include", "sparf_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-fconstexpr-steps=200000000"]


def source_hash():
    """sha1 over the kernel sources + headers + the C ABI header: stamps measured profiles (tests/tools/scale_parity.py) so
    that bench.py can tell when the committed parity numbers were measured on other kernels than the ones it runs"""
    import hashlib
    h = hashlib.sha1()
    for f in sorted(SOURCES) + sorted(HEADERS):
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def _deps(path, seen=None):
    """the files `path` includes with #include "...", transitively (a header change then rebuilds the translation units that
    see it, not all seventeen: mlp_bwd_impl.h is read by two of them, a full rebuild takes six minutes on eight cores)"""
    import re
    seen = set() if seen is None else seen
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path, errors="replace").read(), flags=re.M):
        _deps(os.path.normpath(os.path.join(os.path.dirname(path), inc)), seen)
    return seen


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, out=None, extra_flags=(), tag="", only=None):
    """Compile and link.  `out` / `extra_flags` / `tag` build a variant library next to the
    default one (separate object directory) for A/B measurements via $SPARF_LIB; `only` = the
    translation units the variant's flags affect (the others are linked from the default build)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    bdir = os.path.join(CSRC, "build" + tag)
    out = out or OUT
    os.makedirs(bdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    procs, objs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if only is not None and src not in only:
            objs.append(os.path.join(CSRC, "build", os.path.splitext(src)[0] + ".o"))
            continue
        obj = os.path.join(bdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _newer(obj, sorted(_deps(sp))):
            cmd = [hipcc] + FLAGS + list(extra_flags) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", sp, "-o", obj]
            if verbose:
                print("[sparf_amd.build]", " ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = []
    for src, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            failed.append((src, log))
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(f"--- {s}\n{o}" for s, o in failed))
    if force or procs or _newer(out, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        if verbose:
            print("[sparf_amd.build]", " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
