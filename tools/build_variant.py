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
    # the file lists of THAT revision (translation units get split / renamed between rounds)
    ns = {"__file__": os.path.join(ROOT, "sparf_amd", "build.py"), "__name__": "old_build"}
    exec(subprocess.check_output(["git", "show", f"{rev}:sparf_amd/build.py"], cwd=ROOT).decode(), ns)
    SOURCES, HEADERS, FLAGS = ns["SOURCES"], ns["HEADERS"], ns["FLAGS"]
    files = SOURCES + [h for h in HEADERS if not h.startswith("..")]
    for f in files:
        data = subprocess.check_output(["git", "show", f"{rev}:sparf_amd/csrc/{f}"], cwd=ROOT)
        # api.hip includes "../../include/sparf_hip.h": give the old sources the header of THEIR revision
        data = data.replace(b'"../../include/sparf_hip.h"', b'"sparf_hip.h"')
        open(os.path.join(src_dir, f), "wb").write(data)
    open(os.path.join(src_dir, "sparf_hip.h"), "wb").write(subprocess.check_output(["git", "show", f"{rev}:include/sparf_hip.h"], cwd=ROOT))
    objs, procs = [], []
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    for s in SOURCES:
        obj = os.path.join(src_dir, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        cmd = [hipcc] + FLAGS + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", os.path.join(src_dir, s), "-o", obj]
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
