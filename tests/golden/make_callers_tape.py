"""Generates tests/golden/callers_tape_<settings>.npz FROM THE REFERENCE (run in the build container, where /root/reference
exists; the GPU box only replays the committed files):

    python tests/golden/make_callers_tape.py [--names dtu_nerf dtu_barf llff_sparf replica_sparf] [--rays 4096]

For each of the reference's own settings files behind BASELINE configs 1-4 (nerf_training_w_gt_poses/dtu/nerf.py,
joint_pose_nerf_training/{dtu/barf, llff/sparf, replica/sparf}.py) at BASELINE's sizes -- 4096 rays x (64 + 128) samples -- ONE
training iteration of the reference's unmodified sampler + loss modules (base_losses.py:243-323, corres_loss.py:27-223,
depth_cons_loss.py:31-321) runs on the reference `Graph` (source/models/renderer.py:250-345, 504-593) on the CPU, with every random
draw taken from np.random.RandomState(seed) and the weights from tests/callers_tape.seeded_state; tests/callers_tape.TapedCalls
records every render call.  What is committed per call: the arguments, the outputs the loss code reads, the gradient the loss
sent back into them, the gradient at the pose / pixel inputs; per iteration: the loss terms and the parameter gradients (small
tensors whole, the 256-wide layers as 8192 entries + norm).  ~1-2 MB per settings file, ~10 minutes of CPU for all four.
tests/test_01_reference_tape_gpu.py replays them against the HIP renderer; tests/test_callers_tape_cpu.py checks the tape
machinery itself (a tape replayed on the reference reproduces it exactly).
"""
import argparse
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]

from tests import callers_tape as CT  # noqa: E402

SEEDS = {"dtu_nerf": 11, "dtu_barf": 12, "llff_sparf": 13, "replica_sparf": 14}
# what the loss code reads of a call (`render`'s own all_cumulated is returned and never consumed;
# depth_cons_loss.py:271-273 reads all_cumulated(_fine) of render_to_max and nothing else)
CONSUMED = {"render": ("rgb", "depth", "opacity", "rgb_fine", "depth_fine", "opacity_fine"), "render_to_max": ("all_cumulated", "all_cumulated_fine")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--names", nargs="*", default=list(SEEDS))
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    import torch
    if a.threads:
        torch.set_num_threads(a.threads)
    for name in a.names:
        t0 = time.time()
        tape = CT.record(name, seed=SEEDS[name], rays=a.rays, samples=(64, 128), device="cpu")
        for c in tape["calls"]:
            c["out"] = {k: v for k, v in c["out"].items() if k in CONSUMED[c["method"]]}
        # settings files with density noise (dtu/nerf.py:34): keep the reference's merged fine depths (renderer.py:334-336), 3 MB per
        # render call, so that the replay can render the fine pass AT them (tests/callers_tape.replay(force_fine_depths=True))
        tape["keep_t_fine"] = bool(tape["opt"].get("nerf", {}).get("density_noise_reg"))
        path = CT.save(tape, os.path.join(HERE, f"callers_tape_{name}.npz"))
        print(f"{name}: {len(tape['calls'])} calls {[(c['method'], tuple(c['out'][sorted(c['out'])[0]].shape[:2])) for c in tape['calls']]}, "
              f"losses {tape['losses']}, {os.path.getsize(path) / 1e6:.2f} MB, {time.time() - t0:.0f} s", flush=True)


if __name__ == "__main__":
    main()
