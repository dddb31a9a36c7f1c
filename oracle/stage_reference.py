"""Stage the reference's Python tree into the git-ignored `oracle/_ref/` so that it travels to the GPU box.

TEST INFRASTRUCTURE.  `/root/reference` exists only in the build container; `gpurun` ships the
repository snapshot (built `.so` files and `oracle/_ref/` included -- both are git-ignored, neither is
gpurun-ignored).  This recipe copies, byte for byte and only into `oracle/_ref/`,

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
PARTS = ["source", "train_settings", os.path.join("third_party", "pytorch_ssim"), os.path.join("third_party", "ATE")]


def staged_root():
    """Root to put on sys.path to import the reference: the staged copy if present, else the reference tree itself, else None."""
    if os.path.isdir(os.path.join(DST, "source", "models")):
        return DST
    if os.path.isdir(os.path.join(REF, "source", "models")):
        return REF
    return None


def stage(verbose=True):
    if not os.path.isdir(os.path.join(REF, "source")):
        if verbose:
            print(f"[stage_reference] {REF} absent: nothing staged (using {staged_root() or 'no reference at all'})")
        return staged_root()
    ignore = shutil.ignore_patterns("__pycache__", "*.pyc")
    for part in PARTS:
        src, dst = os.path.join(REF, part), os.path.join(DST, part)
        if not os.path.exists(src):
            continue
        if os.path.isdir(dst):
            shutil.rmtree(dst)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copytree(src, dst, ignore=ignore)
    if verbose:
        n = sum(len(f) for _, _, f in os.walk(DST))
        print(f"[stage_reference] staged {n} files of {REF} into {DST}")
    return DST


if __name__ == "__main__":
    print(stage() or "absent")
    sys.exit(0)
