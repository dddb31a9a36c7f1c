"""Build the kernels of another git revision into sparf_amd/libsparf_hip_<tag>.so so that two
versions can be timed inside ONE gpurun call (box-to-box variance is ~10 %):

    python tools/build_variant.py HEAD old
    SPARF_LIB=sparf_amd/libsparf_hip_old.so python tools/kernel_bench.py bf16

Sources are extracted with `git show` into sparf_amd/csrc_<tag>/ (git-ignored scratch)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparf_amd import build as B                                     # noqa: E402


def main(rev, tag):
    src_dir = os.path.join(ROOT, "sparf_amd", "csrc_" + tag)
    os.makedirs(src_dir, exist_ok=True)
    files = B.SOURCES + [h for h in B.HEADERS if not h.startswith("..")]
    for f in files:
        data = subprocess.check_output(["git", "show", f"{rev}:sparf_amd/csrc/{f}"], cwd=ROOT)
        open(os.path.join(src_dir, f), "wb").write(data)
    # api.hip includes "../../include/sparf_hip.h": csrc_<tag>/ sits at the same depth as csrc/
    objs, procs = [], []
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    for s in B.SOURCES:
        obj = os.path.join(src_dir, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        cmd = [hipcc] + B.FLAGS + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", os.path.join(src_dir, s), "-o", obj]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            raise SystemExit(f"{s}:\n{log}")
    out = os.path.join(ROOT, "sparf_amd", f"libsparf_hip_{tag}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    print(out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
