"""The reference's own training-iteration code against the HIP renderer, TEACHER-FORCED and lease-independent
(VERDICT r04 next-1): the committed tapes tests/golden/callers_tape_<settings>.npz hold, for the reference's `get_config()` of
BASELINE configs 1-4 (nerf_training_w_gt_poses/dtu/nerf.py, joint_pose_nerf_training/{dtu/barf, llff/sparf, replica/sparf}.py) at
BASELINE's sizes (4096 rays x (64 + 128) samples), what ONE iteration of the reference's unmodified sampler and loss modules
(base_losses.py:243-323, corres_loss.py:27-223, depth_cons_loss.py:31-321) asked of the reference `Graph` and what it sent
back: per render call the arguments, the outputs the loss read, the upstream gradient of each output, the gradient at the pose
and pixel inputs; per iteration the gradients of both networks (tests/callers_tape.py; made on the CPU from /root/reference by
tests/golden/make_callers_tape.py).  Here every taped call is replayed on the HIP `Graph` with the SAME arguments, draws and
weights -- the calls whose pixel lists and depth caps derive from earlier renders included (depth_cons_loss.py:254-291) -- and
the taped upstream gradients are pushed back through it.

Nothing in this file depends on the lease: inputs are the committed tape (draws regenerated from its numpy seed and verified
against their checksums, weights from tests/callers_tape.seeded_state), the kernels are deterministic (no atomics).  No reference
code runs here; the file sorts first so that `pytest -x` reaches it (and tests/test_00_scale_gpu.py) whatever happens later.

Bounds: outputs the callers read max|a-b| / max|b| <= 1e-4 in BOTH modes (north_star), no allowance for "derived inputs" any more;
no exception (round 5's NOISY_RESAMPLING is gone: FORCED below).  Gradients: measured values x ~2 (gpurun_out/r06_reference_tape.json,
committed as profiles/r06_reference_tape.json)."""
import json
import os

import pytest
import torch

from tests import callers_tape as CT

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["dtu_nerf", "dtu_barf", "llff_sparf", "replica_sparf"]
#                      outputs   parameter gradients (rel. L2)     per-call input gradients
#                                worst tensor   all parameters     d pose (max-norm)   d pixels (rel. L2)
# measured (profiles/r06_reference_tape.json; deterministic: the same numbers on every lease), maximum over the four settings files
#   fp32               4.5e-6    9.5e-4         3.1e-5             1.2e-3              3.6e-3         (gradients: dtu/nerf.py with forced fine depths;
#   bf16x3             2.2e-5    1.6e-3         5.8e-5             3.3e-3              6.5e-3          the other three 3.1e-4 / 7.8e-4 as in round 5)
BOUNDS = {"fp32": dict(out=2e-5, grad_worst=1.5e-3, grad_all=1e-4, pose=4e-3, pix=1e-2),
          "bf16x3": dict(out=1e-4, grad_worst=2.5e-3, grad_all=3e-4, pose=1e-2, pix=2e-2)}
# dtu/nerf.py:34 adds N(0, 1) noise to the raw density (frequency_nerf.py:191-192): the coarse weights become rough, many pdf bins
# are near-empty, and the inverse-CDF resampling (renderer.py:446-452: (u - cdf_lo) / (cdf_hi - cdf_lo + 1e-8)) moves a fine sample
# by up to a bin width for a 1e-6 change of the coarse weights.  Round 5 compared that settings file's fine pass on two different sample
# sets -- each renderer's own resampling -- under a 2e-3 exception.  Since round 6 its tape holds the reference's merged fine depths
# (renderer.py:334-336) and the call is replayed TWICE:
#   forced   the fine pass rendered AT the taped depths (tests/callers_tape.forced_fine_depths): every key, fine ones included, is held
#            to the common 1e-4 and the fine network's gradient to the common bounds -- no exception left;
#   free     the renderer resamples from its own coarse weights as the product does; what is asserted is statistical -- how many of the
#            786 432 merged samples moved and how far -- because a sample that hops a bin is a different input, not an error.
# Measured (profiles/r06_reference_tape.json): forced, fine outputs 4.0e-6 (fp32) / 5.5e-6 (bf16x3) where round 5's free comparison read
# 3.2e-4 / 6.9e-4; the fine network's worst gradient tensor 9.5e-4 / 1.6e-3 instead of 2.3e-3 / 3.4e-3.
FORCED = {"dtu_nerf"}
# free resampling, per precision: share of the 786 432 merged samples further than 1e-6 of the depth range from the reference's / mean |dt| /
# range / the furthest one (a bin hop: <= a bin width, ~1e-4 of the range) / max-norm error of the fine outputs (round 5's NOISY_RESAMPLING
# quantity).  Bounds = 1.5 x measured: fp32 1.6 % moved, mean 1.0e-7, max 7.0e-5, fine outputs 3.2e-4; bf16x3 5.3 %, 2.5e-7, 9.7e-5, 6.9e-4.
FREE_RESAMPLING = {"fp32": dict(moved=2.4e-2, mean_dt=1.6e-7, max_dt=1.1e-4, out_fine=4.8e-4), "bf16x3": dict(moved=7.9e-2, mean_dt=3.8e-7, max_dt=1.5e-4, out_fine=1.04e-3)}
_REPORT = {}


def tape_path(name):
    return os.path.join(ROOT, "tests", "golden", f"callers_tape_{name}.npz")


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("name", NAMES)
def test_taped_reference_iteration_on_hip_graph(name, precision):
    from sparf_amd.renderer import Graph
    dev = torch.device("cuda:0")
    tape = CT.load(tape_path(name))
    assert tape["rays"] == 4096 and tape["samples"] == (64, 128), "the committed tapes are BASELINE-sized"
    opt = CT.opt_from_json(tape["opt"], precision)
    torch.manual_seed(0)
    graph = Graph(opt, dev)
    graph.train()
    CT.load_seeded(graph, tape["weight_seed"])
    forced = name in FORCED
    if forced:
        assert all(c["t_fine"] is not None for c in tape["calls"] if c["method"] == "render"), "the tape of a density-noise settings file holds the merged fine depths"
        free = CT.replay(tape, graph, opt, dev)                          # the product's own resampling: statistical assertions below
        _REPORT[f"{name}/{precision}/free_resampling"] = free
    r = CT.replay(tape, graph, opt, dev, force_fine_depths=forced)
    _REPORT[f"{name}/{precision}"] = r
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_reference_tape.json"), "w") as f:
        json.dump(_REPORT, f, indent=1, default=str)
    b = BOUNDS[precision]
    assert not r["missing_grads"], r["missing_grads"]
    n_calls = {"dtu_nerf": 1, "dtu_barf": 1}.get(name, 6)           # photometric | + 2 correspondence + 3 depth-consistency renders
    assert len(r["per_call"]) == n_calls
    for i, e in enumerate(r["per_call"]):
        assert not e["_missing_outputs"] and not e["_unused_draws"], (i, e)
        for k, v in e.items():
            if k.startswith("_"):
                continue
            if k == "d_pose":
                bound = b["pose"]
            elif k == "d_pixels":
                bound = b["pix"]
            else:
                bound = b["out"]
            assert v <= bound, (name, precision, "call", i, r["calls"][i], k, v, "bound", bound)
        assert e["_forced"] == (forced and r["calls"][i][0] == "render"), (i, e["_forced"])
    assert r["grad_worst_tensor"] <= b["grad_worst"], (r["grad_worst_name"], r["grad_worst_tensor"])
    assert r["grad_all"] <= b["grad_all"], r["grad_all"]
    assert r["grad_norm_ratio_worst"] <= 1e-2, r["grad_norm_ratio_worst"]      # the whole tensors, where only a subset of the entries is taped
    if forced:
        fb = FREE_RESAMPLING[precision]
        for i, e in enumerate(free["per_call"]):
            tf = e["_t_fine"]
            assert tf["moved_gt_1e6"] <= fb["moved"] and tf["mean"] <= fb["mean_dt"] and tf["max"] <= fb["max_dt"], (name, precision, "free resampling", tf)
            for k, v in e.items():
                if k.startswith("_") or k in ("d_pose", "d_pixels"):
                    continue
                assert v <= (fb["out_fine"] if k.endswith("_fine") else b["out"]), (name, precision, "free", k, v)      # the coarse pass does not depend on the resampling
    kinds = [(m, g) for m, _, g in r["calls"]]
    if n_calls == 6:
        assert kinds == [("render", True)] * 4 + [("render_to_max", False), ("render", True)], kinds
        assert "d_pixels" in r["per_call"][5], "depth_cons_loss.py:291 renders at pixels that carry a gradient"
