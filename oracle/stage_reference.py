"""Stage the reference's Python tree into the git-ignored `oracle/_ref/` so that it travels to the GPU box.

TEST INFRASTRUCTURE.  `/root/reference` exists only in the build container; `gpurun` ships the
repository snapshot (built `.so` files and `oracle/_ref/` included -- both are git-ignored, neither is
gpurun-ignored).  This recipe packs, byte for byte and only into ONE archive `oracle/_ref/reference_tree.zip` (a build artefact like a
compiled oracle/_ref/*.so would be; unpacked into a scratch directory outside the repository when a test imports it),

    source/            the reference package (renderer, NeRF, camera, pose models, LOSS MODULES, ...)
    train_settings/    its settings files (`get_config()` of every BASELINE config)
    third_party/pytorch_ssim, third_party/ATE    the two vendored helpers `source.*` imports (the un-vendored submodules
                       DenseMatching / Hierarchical-Localization are empty in the reference checkout too)

so that on the GPU box
  * `tests/test_reference_callers_gpu.py` can run the reference's own, unmodified loss modules
    (base_losses.py:243-323, corres_loss.py:97-220, depth_cons_loss.py:128-321) once with
    `self.net` = the reference `Graph` (fp32 PyTorch-ROCm ops) and once with `self.net` = the HIP
    `Graph`, and compare every loss term and gradient;
  * `bench.py`'s `cpu_baseline` leg can time the reference module itself on the bench host
    (`kind: "reference"`).
Nothing under `sparf_amd/`, `dropin/` or the timed region of `bench.py` imports `oracle/_ref`; no
reference source enters the git history (`oracle/_ref/` is in `.gitignore`).

    python oracle/stage_reference.py            # idempotent; prints the staged root or "absent"
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SPARF_REFERENCE_ROOT", "/root/reference")
DST = os.path.join(HERE, "_ref")
ARCHIVE = os.path.join(DST, "reference_tree.zip")      # ONE build artefact (like a built .so)
PARTS = ["source", "train_settings", os.path.join("third_party", "pytorch_ssim"), os.path.join("third_party", "ATE")]


def staged_root():
    """What to put on sys.path to import the reference: the staged archive if present, else the reference tree itself, else None."""
    if os.path.isfile(ARCHIVE):
        return ARCHIVE
    if os.path.isdir(os.path.join(REF, "source", "models")):
        return REF
    return None


def import_root():
    """A directory to put on sys.path: the reference tree itself, or the staged archive unpacked into a scratch directory outside
    the repository (zipimport does not resolve the reference's __init__-less sub-packages, e.g. source/utils/geometry)."""
    root = staged_root()
    if root is None or not root.endswith(".zip"):
        return root
    import hashlib
    import tempfile
    import zipfile
    st = os.stat(root)
    tag = hashlib.sha1(f"{root}:{st.st_size}:{st.st_mtime_ns}".encode()).hexdigest()[:12]
    out = os.path.join(tempfile.gettempdir(), f"sparf_reference_{tag}")
    if not os.path.isdir(os.path.join(out, "source", "models")):
        tmp = out + f".{os.getpid()}"
        with zipfile.ZipFile(root) as z:
            z.extractall(tmp)
        try:
            os.replace(tmp, out)
        except OSError:                      # another process got there first
            shutil.rmtree(tmp, ignore_errors=True)
    return out


def read_text(relpath):
    """text of a file of the reference tree (staged archive or the tree itself), e.g. 'source/training/joint_pose_nerf_trainer.py'"""
    root = staged_root()
    if root is None:
        raise FileNotFoundError("no reference tree")
    if root.endswith(".zip"):
        import zipfile
        with zipfile.ZipFile(root) as z:
            return z.read(relpath).decode()
    return open(os.path.join(root, relpath)).read()


def stage(verbose=True):
    if not os.path.isdir(os.path.join(REF, "source")):
        if verbose:
            print(f"[stage_reference] {REF} absent: nothing staged (using {staged_root() or 'no reference at all'})")
        return staged_root()
    import zipfile
    os.makedirs(DST, exist_ok=True)
    for stale in PARTS + ["third_party"]:                       # (earlier revisions of this recipe staged loose files)
        d = os.path.join(DST, stale)
        if os.path.isdir(d):
            shutil.rmtree(d)
    tmp = ARCHIVE + ".tmp"
    n = 0
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        for part in PARTS:
            for dirpath, dirnames, filenames in os.walk(os.path.join(REF, part)):
                dirnames[:] = sorted(d for d in dirnames if d != "__pycache__")
                for f in sorted(filenames):
                    if f.endswith(".pyc"):
                        continue
                    full = os.path.join(dirpath, f)
                    info = zipfile.ZipInfo(os.path.relpath(full, REF), date_time=(2020, 1, 1, 0, 0, 0))     # reproducible archive
                    info.compress_type = zipfile.ZIP_DEFLATED
                    z.writestr(info, open(full, "rb").read())
                    n += 1
    os.replace(tmp, ARCHIVE)
    if verbose:
        print(f"[stage_reference] staged {n} files of {REF} into {ARCHIVE}")
    return ARCHIVE


if __name__ == "__main__":
    print(stage() or "absent")
    sys.exit(0)
