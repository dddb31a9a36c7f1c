"""OPT-IN (needs $SPARF_REFERENCE_ROOT; skipped otherwise -- the lease-independent version of this evidence is
tests/test_01_reference_tape_gpu.py on the committed tapes): the reference's OWN training-iteration code -- `RaySamplingStrategy`,
`define_loss` -> `BasePhotoandReguLoss` (base_losses.py:243-323), the correspondence loss (corres_loss.py:27-223,
base_corres_loss.py:30-375), `DepthConsistencyLoss` (depth_cons_loss.py:31-321), the joint-pose `class Graph(Graph)`
(joint_pose_nerf_trainer.py:710-749) with the reference pose network -- run UNMODIFIED on top of the HIP `Graph`, next to the
same code on top of the reference's `Graph` (fp32 PyTorch-ROCm ops on the same GPU), for the reference's own `get_config()` of
BASELINE configs 1 / 2 / 3 / 4 at BASELINE's sizes: 4096 rays x (64 + 128) samples.

Two comparisons:
  * TEACHER-FORCED (`test_teacher_forced`): the reference run is taped (tests/callers_tape.py) and every render call is
    replayed on the HIP graph with the reference's arguments -- identical inputs for every call, the ones derived from earlier
    renders included -- and the reference's upstream gradients.  Elementwise bounds, the same as for the committed tapes.
  * FREE-RUNNING (`test_free_running`): the chain as the trainer runs it -- the HIP graph's later calls are asked for pixels and depth
    caps derived from ITS OWN earlier outputs (depth_cons_loss.py:199-201, 254-262), data-dependent ray counts may differ by a
    threshold flip -- compared STATISTICALLY: loss terms, gradient distance.  Bounds from the spread over 8 numpy seeds
    (tests/tools/reference_callers_seeds.py -> profiles/r05_reference_callers_seeds.json).
Every draw of both runs comes from np.random.RandomState(seed) (ref_harness.DrawTape(seed=...)): torch's AND numpy's -- round 4
seeded torch only, the loss modules draw from np.random (depth_cons_loss.py:57,181, base_corres_loss.py:164), and the compared
iteration changed from lease to lease."""
import json
import os

import pytest
import torch

from tests import callers_tape as CT
from tests import ref_harness as RH
from tests.test_01_reference_tape_gpu import BOUNDS as TF_BOUNDS
# this opt-in comparison lets BOTH renderers resample their own fine depths (no forced replay as in test_01): round 5's exception for the
# density-noise settings file applies here (the reference against itself, GPU vs CPU, differs by 3.7e-4 on these keys)
NOISY_RESAMPLING = {"dtu_nerf": {"fp32": 1e-3, "bf16x3": 2e-3}}
NOISY_GRAD_WORST = {"dtu_nerf": 7e-3}

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not os.environ.get("SPARF_REFERENCE_ROOT"),
                                 reason="opt-in: set $SPARF_REFERENCE_ROOT to a reference checkout or an archive of oracle/stage_reference.py "
                                        "(the committed-tape version of this test is tests/test_01_reference_tape_gpu.py)")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ITER = 110000            # past every start gate of the three settings files; c2f progress 0.55 of [0.4, 0.7]
SEED = 3
# free-running chain: loss terms (relative), parameter gradients (relative L2: worst tensor / all parameters), pose-network
# gradient (max-norm relative).  Worst over 8 numpy seeds x the two SPARF settings + 2 seeds x the two DTU ones
# (profiles/r05_reference_callers_seeds.json; no ray count differed between the two renderers in any of the 20 iterations):
#              loss     worst tensor (dtu_nerf)   all parameters   pose
#   fp32       1.5e-7   1.8e-4 (9.8e-4)           5.2e-5           1.9e-3
#   bf16x3     1.8e-5   8.4e-4 (3.2e-3)           1.7e-4           2.6e-3           bounds = ~3x
FREE = {"fp32": dict(loss=1e-5, grad_worst=6e-4, grad_all=1.5e-4, pose=6e-3),
        "bf16x3": dict(loss=6e-5, grad_worst=2.5e-3, grad_all=5e-4, pose=1e-2)}
# teacher-forced outputs: the reference runs on the GPU here (rocBLAS summation order) and is itself up to 5e-5 from the CPU reference
# the committed tapes hold (HIP fp32 against those: 4.5e-6) -- both modes are held to the north_star's 1e-4
LIVE_OUT = 1e-4
_REPORT = {}


def _dump(key, value):
    _REPORT[key] = value
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r05_reference_callers.json"), "w") as f:
        json.dump(_REPORT, f, indent=1, default=str)


def teacher_forced(name, precision, seed=SEED):
    from sparf_amd.renderer import Graph
    dev = torch.device("cuda:0")
    tape = CT.record(name, seed=seed, rays=4096, samples=(64, 128), device=dev, iteration=ITER)      # the reference Graph on this GPU
    torch.cuda.empty_cache()
    opt = CT.opt_from_json(tape["opt"], precision)
    graph = Graph(opt, dev)
    graph.train()
    CT.load_seeded(graph, tape["weight_seed"])
    return CT.replay(tape, graph, opt, dev)


def free_running(name, precision, seed=SEED, bare_cuda=False):
    dev = "cuda:0"
    opt = RH.load_settings(name, rays=4096, samples=(64, 128))
    scene = RH.make_scene(name, opt, dev)
    torch.manual_seed(0)
    g_ref, o_ref = RH.build_graph("reference", opt, scene, dev)
    CT.load_seeded(g_ref, 1000)
    state = {k: v.detach().clone() for k, v in g_ref.state_dict().items()}
    tape = RH.DrawTape(seed=seed)
    r_ref = RH.training_iteration(g_ref, o_ref, scene, ITER, tape, "record")
    del g_ref
    torch.cuda.empty_cache()
    # (nerf_trainer.py:112-114 hands Graph the trainer's device, a bare "cuda": covered by the bare_cuda case)
    g_hip, o_hip = RH.build_graph("hip", opt, scene, "cuda" if bare_cuda else dev, state=state, precision=precision)
    r_hip = RH.training_iteration(g_hip, o_hip, scene, ITER, tape, "replay")
    c = RH.compare(r_ref, r_hip)
    c["leftover_draws"] = {str(k): v for k, v in tape.leftover().items()}
    c["resized_draws"] = [str(x) for x in tape.resized]
    return c


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("name", ["dtu_nerf", "dtu_barf", "llff_sparf", "replica_sparf"])
def test_teacher_forced(name, precision):
    r = teacher_forced(name, precision)
    _dump(f"teacher_forced/{name}/{precision}", r)
    b = TF_BOUNDS[precision]
    assert not r["missing_grads"], r["missing_grads"]
    for i, e in enumerate(r["per_call"]):
        assert not e["_missing_outputs"] and not e["_unused_draws"], (i, e)
        is_tomax = r["calls"][i][0] == "render_to_max"
        for k, v in e.items():
            if k.startswith("_"):
                continue
            # what the callers read: rgb / depth / opacity of `render`, all_cumulated(_fine) of `render_to_max` (depth_cons_loss.py:271-273);
            # `render`'s own all_cumulated (transmittance before the last sample) is returned and never consumed
            consumed = k.startswith("all_cumulated") == is_tomax or k in ("d_pose", "d_pixels")
            bound = b["pose"] if k == "d_pose" else b["pix"] if k == "d_pixels" else LIVE_OUT if consumed else 20 * LIVE_OUT
            if k.endswith("_fine") and name in NOISY_RESAMPLING and not is_tomax:
                bound = NOISY_RESAMPLING[name][precision]
            assert v <= bound, (name, precision, "call", i, r["calls"][i], k, v, "bound", bound)
    assert r["grad_worst_tensor"] <= NOISY_GRAD_WORST.get(name, b["grad_worst"]), (r["grad_worst_name"], r["grad_worst_tensor"])
    assert r["grad_all"] <= b["grad_all"], r["grad_all"]


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("name", ["dtu_nerf", "dtu_barf", "llff_sparf", "replica_sparf"])
def test_free_running(name, precision):
    c = free_running(name, precision, bare_cuda=(name == "dtu_barf"))
    _dump(f"free_running/{name}/{precision}", c)
    b = FREE[precision]
    assert not c["missing_grads"], c["missing_grads"]
    ref_c, hip_c = c["calls"]["ref"], c["calls"]["test"]
    # the photometric and correspondence renders are asked for the same ray sets by construction; the two last depth-consistency
    # renders take the rays whose reprojection fell inside the image (depth_cons_loss.py:254-256) and whose visibility passed
    # `>= 0.2` (:274): thresholds a ray can sit on to within the renderers' 1e-5
    assert len(ref_c) == len(hip_c) and ref_c[:4] == hip_c[:4], ("the two renderers were asked for different ray sets", c["calls"])
    assert all(abs(ref_c[i][1] - hip_c[i][1]) <= 8 for i in range(4, len(ref_c))), ("more than threshold flips", c["calls"])
    for k, v in c["loss"].items():
        assert v["rel"] <= b["loss"], (name, precision, "loss term", k, v)
    assert c["grad_worst_tensor"] <= NOISY_GRAD_WORST.get(name, b["grad_worst"]), (c["grad_worst_name"], c["grad_worst_tensor"])
    assert c["grad_all"] <= b["grad_all"], c["grad_all"]
    if name == "dtu_nerf":
        assert c["grad_pose"] is None                       # fixed GT poses: the plain Graph, no pose network
    else:
        assert c["grad_pose"] is not None and c["grad_pose"] <= b["pose"], c["grad_pose"]
