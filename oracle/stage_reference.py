"""Where the tests find the reference's Python tree, and an OPT-IN tool that packs it into one archive.

TEST INFRASTRUCTURE.  Nothing under `sparf_amd/`, `dropin/`, `__graft_entry__.build()` or the default `bench.py` run
imports, stages or executes the reference.  The reference tree is looked for, in this order, at

    $SPARF_REFERENCE_ROOT    a checkout of google-research/sparf, or an archive made by this tool (path relative to the
                             repository root or absolute)
    /root/reference          the read-only checkout of the build container

and nowhere else: in particular NOT implicitly inside this repository (round 4's `build()` packed the tree into
`oracle/_ref/` on every build, so the untrusted sources travelled inside every snapshot and `bench.py` executed them by
default).  What needs the reference ON THE GPU BOX is now served by committed fixtures instead: `tests/golden/callers_tape_*.npz`
hold what the reference's own loss modules asked of the renderer and got back (made here, on the CPU, by
`tests/golden/make_callers_tape.py`; replayed against the HIP renderer by `tests/test_01_reference_tape_gpu.py`).

The live comparison (`tests/test_reference_callers_gpu.py`: reference `Graph` and HIP `Graph` side by side on one GPU
under the reference's unmodified loss code) is opt-in: it runs where `$SPARF_REFERENCE_ROOT` is set, e.g. in a builder's own
`gpurun` session after

    python oracle/stage_reference.py --out oracle/_ref/reference_tree.zip      # explicit target; oracle/_ref is git-ignored
    SPARF_REFERENCE_ROOT=oracle/_ref/reference_tree.zip python -m pytest tests/test_reference_callers_gpu.py -m gpu

The archive holds, byte for byte: source/, train_settings/, third_party/pytorch_ssim, third_party/ATE (the two vendored
helpers `source.*` imports; DenseMatching / Hierarchical-Localization are empty submodules in the reference checkout too).
It is unpacked, by the process that imports it, into a directory of mode 0700 owned by the current user.
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CONTAINER_REF = "/root/reference"
PARTS = ["source", "train_settings", os.path.join("third_party", "pytorch_ssim"), os.path.join("third_party", "ATE")]


def _is_tree(path):
    return os.path.isdir(os.path.join(path, "source", "models"))


def staged_root():
    """The reference tree to import from: $SPARF_REFERENCE_ROOT (a checkout or an archive of this tool), else the build
    container's /root/reference, else None.  Never a path chosen implicitly inside the repository."""
    env = os.environ.get("SPARF_REFERENCE_ROOT")
    if env:
        path = env if os.path.isabs(env) else os.path.join(ROOT, env)
        if os.path.isfile(path) and path.endswith(".zip"):
            return path
        if _is_tree(path):
            return path
        raise FileNotFoundError(f"$SPARF_REFERENCE_ROOT={env!r}: neither a reference checkout (source/models/) nor an archive made by "
                                "oracle/stage_reference.py")
    if _is_tree(CONTAINER_REF):
        return CONTAINER_REF
    return None


def _private_cache_dir():
    """a directory only the current user can write to (mode 0700, ownership verified): where an archive is unpacked"""
    base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
    d = os.path.join(base, "sparf_amd_reference")
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.stat(d)
    if st.st_uid != os.getuid() or (st.st_mode & 0o077):
        # not ours, or open to others: do not trust whatever is in there
        import tempfile
        return tempfile.mkdtemp(prefix="sparf_amd_reference_")          # mode 0700, fresh
    return d


def import_root():
    """A directory to put on sys.path: the reference checkout itself, or the archive unpacked into a private (0700, owned by
    this user) directory outside the repository (zipimport does not resolve the reference's __init__-less sub-packages)."""
    root = staged_root()
    if root is None or not root.endswith(".zip"):
        return root
    import hashlib
    import zipfile
    h = hashlib.sha256()
    with open(root, "rb") as f:
        h.update(f.read())
    out = os.path.join(_private_cache_dir(), h.hexdigest()[:16])          # keyed on the archive's CONTENT
    if not _is_tree(out):
        tmp = out + f".{os.getpid()}"
        with zipfile.ZipFile(root) as z:
            for info in z.infolist():                 # no absolute paths, no climbing out of the target
                name = info.filename
                if name.startswith(("/", "\\")) or ".." in name.replace("\\", "/").split("/"):
                    raise ValueError(f"refusing archive member {name!r}")
            z.extractall(tmp)
        try:
            os.replace(tmp, out)
        except OSError:                      # another process of this user got there first
            shutil.rmtree(tmp, ignore_errors=True)
    return out


def read_text(relpath):
    """text of a file of the reference tree (archive or checkout), e.g. 'source/training/joint_pose_nerf_trainer.py'"""
    root = staged_root()
    if root is None:
        raise FileNotFoundError("no reference tree")
    if root.endswith(".zip"):
        import zipfile
        with zipfile.ZipFile(root) as z:
            return z.read(relpath).decode()
    return open(os.path.join(root, relpath)).read()


def stage(out, src=None, verbose=True):
    """Pack the reference tree `src` (default: /root/reference) into the archive `out`.  Explicit opt-in: no default target."""
    import zipfile
    src = src or CONTAINER_REF
    if not _is_tree(src):
        raise FileNotFoundError(f"{src}: not a reference checkout")
    out = out if os.path.isabs(out) else os.path.join(os.getcwd(), out)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    tmp = out + ".tmp"
    n = 0
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        for part in PARTS:
            for dirpath, dirnames, filenames in os.walk(os.path.join(src, part)):
                dirnames[:] = sorted(d for d in dirnames if d != "__pycache__")
                for f in sorted(filenames):
                    if f.endswith(".pyc"):
                        continue
                    full = os.path.join(dirpath, f)
                    info = zipfile.ZipInfo(os.path.relpath(full, src), date_time=(2020, 1, 1, 0, 0, 0))     # reproducible archive
                    info.compress_type = zipfile.ZIP_DEFLATED
                    z.writestr(info, open(full, "rb").read())
                    n += 1
    os.replace(tmp, out)
    if verbose:
        print(f"[stage_reference] packed {n} files of {src} into {out}")
    return out


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", required=True, help="archive to write (explicit; e.g. oracle/_ref/reference_tree.zip, git-ignored)")
    ap.add_argument("--src", default=None, help=f"reference checkout (default {CONTAINER_REF})")
    a = ap.parse_args()
    print(stage(a.out, a.src))
    sys.exit(0)
