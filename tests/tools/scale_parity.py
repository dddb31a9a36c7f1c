"""Measure parity at the BASELINE config shapes (tests/scale_cases.py) and write the numbers to
a JSON under gpurun_out/ (copy to profiles/rNN_parity_scale.json; bench.py reports them).

    python tests/tools/scale_parity.py [--configs 1,2,3,4] [--precisions fp32,bf16x3,bf16] [--yardstick] [--out gpurun_out/parity_scale.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="1,2,3,4")
ap.add_argument("--precisions", default="fp32,bf16x3,bf16")
ap.add_argument("--yardstick", action="store_true", help="also measure the fp32 reference's own distance to the float64 referee")
ap.add_argument("--rays-scale", type=float, default=1.0)
ap.add_argument("--seeds", default="0", help="comma-separated seeds of the random draws (rays, jitter, grid, noise, loss functional)")
ap.add_argument("--referee-device", default="cpu", help="cpu (the oracle as pinned) or cuda:0 (same float64 PyTorch code through PyTorch-ROCm kernels)")
ap.add_argument("--threads", type=int, default=32, help="torch CPU threads for the referee")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_scale.json"))
args = ap.parse_args()

import torch
from tests import scale_cases as S
torch.set_num_threads(args.threads)

results = []
for cfg in [int(c) for c in args.configs.split(",")]:
  for seed in [int(x) for x in args.seeds.split(",")]:
    for prec in args.precisions.split(","):
        r = S.run_case(cfg, prec, yardstick=args.yardstick and prec == "fp32", rays_scale=args.rays_scale, referee_device=args.referee_device, seed=seed)
        r["seed"] = seed
        results.append(r)
        e = r["hip"]
        line = dict(config=cfg, precision=prec, seed=seed, rays=r["rays"], outputs_worst=e["outputs_worst"], rendered_worst=e["rendered_worst"], grad_l2_worst=e["param_grad_rel_l2_worst"],
                    grad_l2_all=e["param_grad_rel_l2_all"], grad_maxrel_worst=e["param_grad_maxrel_worst"],
                    d_origins=e.get("d_origins_rel_l2"), d_viewdirs=e.get("d_viewdirs_rel_l2"), d_pose=e.get("d_pose_maxrel"),
                    t_exact=r["t_coarse_bit_exact"], t_fine=r["t_fine_vs_sampler_oracle_maxabs"], to_max=r.get("to_max"),
                    referee_s=r["referee_seconds"])
        if "reference_fp32" in r:
            y = r["reference_fp32"]
            line["reference_fp32_vs_referee"] = dict(outputs_worst=y["outputs_worst"], grad_l2_worst=y["param_grad_rel_l2_worst"],
                                                     d_origins=y.get("d_origins_rel_l2"))
        print(json.dumps(line), flush=True)
summary = {}
for r in results:
    e = r["hip"]
    kind = "inverse_depth" if r["config"] == 3 else "metric_depth"
    s = summary.setdefault(r["precision"], {}).setdefault(kind, dict(rendered_outputs_max_rel=0.0, per_sample_outputs_max_rel=0.0, param_grad_rel_l2_worst_tensor=0.0,
                                                                       param_grad_rel_l2_all=0.0, ray_grad_rel_l2=0.0, pose_grad_max_rel=0.0, configs=[]))
    s["rendered_outputs_max_rel"] = max(s["rendered_outputs_max_rel"], e["rendered_worst"])
    s["per_sample_outputs_max_rel"] = max(s["per_sample_outputs_max_rel"], e["per_sample_worst"])
    s["param_grad_rel_l2_worst_tensor"] = max(s["param_grad_rel_l2_worst_tensor"], e["param_grad_rel_l2_worst"])
    s["param_grad_rel_l2_all"] = max(s["param_grad_rel_l2_all"], e["param_grad_rel_l2_all"])
    s["ray_grad_rel_l2"] = max(s["ray_grad_rel_l2"], e.get("d_origins_rel_l2", 0.0), e.get("d_viewdirs_rel_l2", 0.0))
    s["pose_grad_max_rel"] = max(s["pose_grad_max_rel"], e.get("d_pose_maxrel", 0.0))
    if r["config"] not in s["configs"]:
        s["configs"].append(r["config"])
    if "reference_fp32" in r:           # yardstick: the fp32 reference's own distance to the float64 referee on the same inputs
        y = r["reference_fp32"]
        ys = summary.setdefault("reference_fp32", {}).setdefault(kind, dict(rendered_outputs_max_rel=0.0, per_sample_outputs_max_rel=0.0,
                                                                             param_grad_rel_l2_worst_tensor=0.0, param_grad_rel_l2_all=0.0, ray_grad_rel_l2=0.0, configs=[]))
        ys["rendered_outputs_max_rel"] = max(ys["rendered_outputs_max_rel"], y["rendered_worst"])
        ys["per_sample_outputs_max_rel"] = max(ys["per_sample_outputs_max_rel"], y["per_sample_worst"])
        ys["param_grad_rel_l2_worst_tensor"] = max(ys["param_grad_rel_l2_worst_tensor"], y["param_grad_rel_l2_worst"])
        ys["param_grad_rel_l2_all"] = max(ys["param_grad_rel_l2_all"], y["param_grad_rel_l2_all"])
        ys["ray_grad_rel_l2"] = max(ys["ray_grad_rel_l2"], y.get("d_origins_rel_l2", 0.0), y.get("d_viewdirs_rel_l2", 0.0))
        if r["config"] not in ys["configs"]:
            ys["configs"].append(r["config"])
os.makedirs(os.path.dirname(args.out), exist_ok=True)
from sparf_amd.build import source_hash
json.dump(dict(_meta=dict(what="HIP path vs float64 referee at BASELINE config shapes (tests/scale_cases.py)", lib=os.environ.get("SPARF_LIB", "default"),
                          kernel_source_hash=source_hash()),
               summary=summary, cases=results), open(args.out, "w"), indent=1)
print("wrote", args.out)
