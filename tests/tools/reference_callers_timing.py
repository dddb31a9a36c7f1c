"""What the drop-in buys an UNMODIFIED trainer: wall time of the reference's own training-iteration code
(tests/ref_harness.training_iteration: RaySamplingStrategy -> Graph -> define_loss -> compute_loss -> backward; the optimiser step
is not part of it) with `self.net` = the reference `Graph` as PyTorch-ROCm ops on the MI355X, and with `self.net` = the HIP `Graph`
in its default mode and in fp32 mode.  BASELINE configs 1-4 at 4096 rays x (64 + 128); loss modules built once, N iterations
timed after warm-up (torch.cuda.synchronize on both sides).  The reference's CPU number is bench.py's `cpu_baseline`.

    python tests/tools/reference_callers_timing.py [--iters 10] [--out gpurun_out/r04_reference_callers_timing.json]
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
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04_reference_callers_timing.json"))
args = ap.parse_args()
assert RH.install_reference(), "needs the reference tree: set $SPARF_REFERENCE_ROOT (oracle/stage_reference.py)"
from easydict import EasyDict as edict
from source.training.core.loss_factory import define_loss
from source.training.core.sampling_strategies import RaySamplingStrategy

dev = "cuda:0"
doc = {}
for name in args.settings.split(","):
    row = {}
    for kind, prec in (("reference", None), ("hip", None), ("hip", "fp32")):
        opt = RH.load_settings(name, rays=args.rays, samples=(64, 128))
        scene = RH.make_scene(name, opt, dev)
        torch.manual_seed(0)
        graph, o = RH.build_graph(kind, opt, scene, dev, precision=prec)
        loss_module = define_loss(o.loss_type, o, graph, scene.train_data, dev, flow_net=scene.flow_net)
        sampler = RaySamplingStrategy(o, data_dict=scene.train_data.all, device=dev)

        def one(it):
            data_dict = edict(scene.train_data.all)
            data_dict.iter = it
            if o.barf_c2f is not None and o.apply_cf_pe:
                graph.nerf.progress.data.fill_(it / o.max_iter)
                graph.nerf_fine.progress.data.fill_(it / o.max_iter)
            rays = sampler(o.nerf.rand_rays, sample_in_center=False)
            out = graph.render_image_at_specific_rays(o, data_dict, ray_idx=rays, iter=it, mode="train")
            data_dict.poses_w2c = graph.get_w2c_pose(o, data_dict, mode="train")
            loss, _, _ = loss_module.compute_loss(o, data_dict, out, mode="train", plot=False, iteration=it)
            graph.zero_grad(set_to_none=True)
            loss["all"].backward()
            return float(loss["all"].detach())

        for i in range(3):
            one(110000 + i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.iters):
            one(110010 + i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.iters * 1e3
        from sparf_amd.frequency_nerf import DEFAULT_PRECISION
        row[kind if kind == "reference" else f"hip_{prec or DEFAULT_PRECISION}"] = round(ms, 2)
        del graph, loss_module
        torch.cuda.empty_cache()
    row["speedup_default"] = round(row["reference"] / row[[k for k in row if k.startswith("hip_") and k != "hip_fp32"][0]], 2)
    doc[name] = row
    print(name, json.dumps(row), flush=True)
os.makedirs(os.path.dirname(args.out), exist_ok=True)
json.dump(dict(_meta=dict(what="ms per training iteration (reference's own sampler + loss modules + backward, no optimiser step) with self.net = the reference "
                               "Graph as PyTorch-ROCm ops on the GPU vs self.net = the HIP Graph", rays=args.rays, samples="64+128", iters=args.iters), **doc),
          open(args.out, "w"), indent=1)
print("wrote", args.out)
