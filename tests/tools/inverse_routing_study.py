"""Which sample rows of an INVERSE-depth pass need fp32 arithmetic for the rendered outputs to hold 1e-4?  

Inverse-depth sampling (renderer.py:413-416) puts sample i of a ray at t = 1 / (1 - (u + i) / N + 1e-8): only the LAST stratified
sample leaves [1, N] (t = N / (1 - u), up to 1e8).  This tool EMULATES row-level routing before any kernel implements it: the
same passes (BASELINE config 3 shape: 2048 rays on pixel lists, 64 coarse + 128 fine samples, six seeds) run once on the HIP
bf16x3 kernels and once on the HIP fp32 kernels, from identical rays / depth samples; the per-sample outputs
(density_samples, rgb_samples) are then mixed row by row -- fp32 where t > threshold, bf16x3 elsewhere -- composited by the
oracle in float64 (oracle.composite) and compared with the float64 referee (oracle.pass_fixed) on the same inputs.
Output: rendered-output error (rgb, depth, opacity, weights, depth_var: max|a-b| / max|b|) per threshold, per pass, per seed.

    python tests/tools/inverse_routing_study.py [--seeds 0,1,2,3,4,5] [--out gpurun_out/r04_inverse_routing_study.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]

import numpy as np
import torch

from oracle import nerf_oracle as O
from sparf_amd.renderer import Graph
from sparf_amd import ops
from tests import scale_cases as S
from tests.golden.recipe import make_state_dict, ring_cameras

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", default="0,1,2,3,4,5")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04_inverse_routing_study.json"))
args = ap.parse_args()

dev = torch.device("cuda:0")
cfg = S.CONFIGS[3]
B, R, H, W, nc, nf = 1, cfg["R"], cfg["H"], cfg["W"], 64, 128
RENDERED = ("rgb", "depth", "opacity", "weights", "depth_var")
THRESHOLDS = [float("inf"), 64.0, 32.0, 16.0, 8.0, 0.0]          # inf: all rows bf16x3; 64: the last coarse sample only; 0: all rows fp32
results = []
for seed in [int(s) for s in args.seeds.split(",")]:
    opts = {p: S.case_opt(cfg, p) for p in ("fp32", "bf16x3!")}
    graphs = {}
    for p, o in opts.items():
        g = Graph(o, dev)
        g.nerf.load_state_dict(make_state_dict(o, 103, None))
        g.nerf_fine.load_state_dict(make_state_dict(o, 203, None))
        graphs[p] = g
    pose, intr = ring_cameras(1, seed=3, H=H, W=W, f=cfg["f"])
    rs = np.random.RandomState(3000 + seed)
    pixels = torch.from_numpy(rs.uniform(0, [W - 1, H - 1], size=(R, 2)).astype(np.float32)).to(dev)
    jitter = torch.from_numpy(rs.uniform(size=(B, R, nc, 1)).astype(np.float32))
    grid = torch.from_numpy(rs.uniform(size=nf + 1).astype(np.float32))
    with torch.no_grad():
        center, ray = ops.ray_gen(pose.to(dev), intr.to(dev), pixels=pixels)
        o32 = opts["fp32"]
        t_c = O.sample_depth(o32, B, R, nc, cfg["rng"], "train", jitter).to(dev)
        per = {}
        for p in ("fp32", "bf16x3!"):
            per[p] = dict(coarse=graphs[p].nerf.render_pass(opts[p], center, ray, t_c, mode="train"))
        # merged fine depths: from the fp32 run's coarse weights (the same t for both precisions)
        u_mid = 0.5 * (grid[:-1] + grid[1:]).to(dev)
        merged, _ = ops.sample_fine(per["fp32"]["coarse"]["weights"].reshape(B * R, nc), t_c.reshape(B * R, nc), u_mid, 1.0, 0.0)
        t_f = merged.view(B, R, nc + nf, 1)
        for p in ("fp32", "bf16x3!"):
            per[p]["fine"] = graphs[p].nerf_fine.render_pass(opts[p], center, ray, t_f, mode="train")
        sd = dict(coarse=graphs["fp32"].nerf.state_dict(), fine=graphs["fp32"].nerf_fine.state_dict())
        row = dict(seed=seed, t_max=float(t_c.max()), n_rows_t_gt_64=dict(coarse=int((t_c > 64).sum()), fine=int((t_f > 64).sum())), passes={})
        for name, t in (("coarse", t_c), ("fine", t_f)):
            pc = {k: v.detach().to(dev) for k, v in sd[name].items()}
            # the referee of tests/scale_cases.py: fp32 points and encoding arguments, float64 downstream
            ref = O.pass_fixed(o32, pc, center, ray, t, mode="train", fine=(name == "fine"), compute_dtype=torch.float64)
            errs = {}
            for thr in THRESHOLDS:
                far = (t[..., 0] > thr)
                dens = torch.where(far, per["fp32"][name]["density_samples"], per["bf16x3!"][name]["density_samples"])
                rgbs = torch.where(far[..., None], per["fp32"][name]["rgb_samples"], per["bf16x3!"][name]["rgb_samples"])
                comp = O.composite(o32, ray.double(), rgbs.double(), dens.double(), t.double())
                e = {k: S.max_rel(comp[k], ref[k]) for k in RENDERED}
                e["worst"] = max(e.values())
                e["rows_fp32"] = int(far.sum())
                errs[str(thr)] = e
            # the HIP kernels' own rendered outputs (their fp32 compositing) for reference
            for p in ("fp32", "bf16x3!"):
                errs["hip_" + p] = {k: S.max_rel(per[p][name][k].reshape(ref[k].shape), ref[k]) for k in RENDERED}
                errs["hip_" + p]["worst"] = max(errs["hip_" + p].values())
            row["passes"][name] = errs
        results.append(row)
        print(json.dumps(dict(seed=seed, t_max=row["t_max"], far=row["n_rows_t_gt_64"],
                              coarse={k: f"{v['worst']:.1e}" for k, v in row["passes"]["coarse"].items()},
                              fine={k: f"{v['worst']:.1e}" for k, v in row["passes"]["fine"].items()})), flush=True)
    del graphs
    torch.cuda.empty_cache()
summary = {}
for name in ("coarse", "fine"):
    for k in results[0]["passes"][name]:
        summary.setdefault(name, {})[k] = max(r["passes"][name][k]["worst"] for r in results)
os.makedirs(os.path.dirname(args.out), exist_ok=True)
json.dump(dict(_meta=dict(what="rendered-output error of an inverse-depth pass when rows with t > threshold take the fp32 kernels' per-sample outputs and the "
                               "others the bf16x3 kernels' (emulated routing; composite + referee in float64)", thresholds=[str(t) for t in THRESHOLDS]),
               summary_worst_over_seeds=summary, seeds=results), open(args.out, "w"), indent=1)
print(json.dumps(summary))
print("wrote", args.out)
