"""How far is the fp32 REFERENCE from itself when only the order of its fp32 summations changes?  The yardstick of
tests/test_reference_callers_gpu.py.

The same training iteration of the reference's own code (tests/ref_harness.py: RaySamplingStrategy -> Graph -> define_loss ->
backward; BASELINE configs 2 / 3 / 4 at 4096 rays x (64 + 128)), same weights, same random draws (DrawTape), run twice with the
reference's own `Graph`: once as fp32 PyTorch-ROCm ops on the GPU (rocBLAS GEMMs), once as fp32 PyTorch ops on the host CPU
(another GEMM blocking, another reduction order).  Both are "the reference in fp32"; their mutual distance -- loss terms, every
render call's outputs, the gradients of both networks and of the pose network -- is what no fp32 implementation can be expected to
undercut against either of them.  Written to gpurun_out/r04_reference_callers_yardstick.json (committed under profiles/).

    python tests/tools/reference_callers_yardstick.py [--settings dtu_barf,llff_sparf,replica_sparf] [--rays 4096]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]

import torch

from tests import ref_harness as RH

ap = argparse.ArgumentParser()
ap.add_argument("--settings", default="dtu_nerf,dtu_barf,llff_sparf,replica_sparf")
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--iter", type=int, default=110000)
ap.add_argument("--threads", type=int, default=32)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04_reference_callers_yardstick.json"))
args = ap.parse_args()
torch.set_num_threads(args.threads)
assert RH.install_reference(), "needs the reference tree: set $SPARF_REFERENCE_ROOT (oracle/stage_reference.py)"

doc = {}
for name in args.settings.split(","):
    t0 = time.time()
    opt = RH.load_settings(name, rays=args.rays, samples=(64, 128))
    scene = RH.make_scene(name, opt, "cuda:0")
    torch.manual_seed(0)
    g_gpu, o_gpu = RH.build_graph("reference", opt, scene, "cuda:0")
    state = {k: v.detach().cpu().clone() for k, v in g_gpu.state_dict().items()}
    tape = RH.DrawTape()
    r_gpu = RH.training_iteration(g_gpu, o_gpu, scene, args.iter, tape, "record")
    del g_gpu
    torch.cuda.empty_cache()
    t1 = time.time()
    opt_c = RH.load_settings(name, rays=args.rays, samples=(64, 128))
    opt_c.device = "cpu"
    scene_c = RH.make_scene(name, opt_c, "cpu")
    g_cpu, o_cpu = RH.build_graph("reference", opt_c, scene_c, "cpu", state=state)
    r_cpu = RH.training_iteration(g_cpu, o_cpu, scene_c, args.iter, tape, "replay")
    c = RH.compare(r_gpu, r_cpu)
    c["seconds"] = dict(gpu=round(t1 - t0, 1), cpu=round(time.time() - t1, 1))
    c["resized_draws"] = [str(x) for x in tape.resized]
    doc[name] = c
    print(name, json.dumps(dict(loss={k: v["rel"] for k, v in c["loss"].items() if not k.endswith("_after_w")}, grad_worst=c["grad_worst_tensor"],
                                grad_worst_name=c["grad_worst_name"], grad_all=c["grad_all"], grad_pose=c["grad_pose"],
                                same_calls=c["calls"]["ref"] == c["calls"]["test"], seconds=c["seconds"])), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(dict(_meta=dict(what="reference Graph fp32 on the GPU (PyTorch-ROCm) vs the same reference Graph fp32 on the host CPU: identical weights, "
                                   "rays and draws; the fp32 reference's distance to itself under another summation order", rays=args.rays, iteration=args.iter),
                   **doc), open(args.out, "w"), indent=1, default=str)
print("wrote", args.out)
