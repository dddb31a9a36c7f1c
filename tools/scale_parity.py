"""Measure parity at the BASELINE config shapes (tests/scale_cases.py) and write the numbers to
a JSON under gpurun_out/ (copy to profiles/rNN_parity_scale.json; bench.py reports them).

    python tools/scale_parity.py [--configs 1,2,3,4] [--precisions fp32,bf16x3,bf16] [--yardstick] [--out gpurun_out/parity_scale.json]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="1,2,3,4")
ap.add_argument("--precisions", default="fp32,bf16x3,bf16")
ap.add_argument("--yardstick", action="store_true", help="also measure the fp32 reference's own distance to the float64 referee")
ap.add_argument("--rays-scale", type=float, default=1.0)
ap.add_argument("--referee-device", default="cpu", help="cpu (the oracle as pinned) or cuda:0 (same float64 PyTorch code through PyTorch-ROCm kernels)")
ap.add_argument("--threads", type=int, default=32, help="torch CPU threads for the referee")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_scale.json"))
args = ap.parse_args()

import torch
from tests import scale_cases as S
torch.set_num_threads(args.threads)

results = []
for cfg in [int(c) for c in args.configs.split(",")]:
    for prec in args.precisions.split(","):
        r = S.run_case(cfg, prec, yardstick=args.yardstick and prec == "fp32", rays_scale=args.rays_scale, referee_device=args.referee_device)
        results.append(r)
        e = r["hip"]
        line = dict(config=cfg, precision=prec, rays=r["rays"], outputs_worst=e["outputs_worst"], grad_l2_worst=e["param_grad_rel_l2_worst"],
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
    s = summary.setdefault(r["precision"], dict(outputs_worst=0.0, param_grad_rel_l2_worst=0.0, param_grad_rel_l2_all_worst=0.0, ray_grad_rel_l2_worst=0.0, configs=[]))
    s["outputs_worst"] = max(s["outputs_worst"], e["outputs_worst"])
    s["param_grad_rel_l2_worst"] = max(s["param_grad_rel_l2_worst"], e["param_grad_rel_l2_worst"])
    s["param_grad_rel_l2_all_worst"] = max(s["param_grad_rel_l2_all_worst"], e["param_grad_rel_l2_all"])
    s["ray_grad_rel_l2_worst"] = max(s["ray_grad_rel_l2_worst"], e.get("d_origins_rel_l2", 0.0), e.get("d_viewdirs_rel_l2", 0.0))
    s["configs"].append(r["config"])
os.makedirs(os.path.dirname(args.out), exist_ok=True)
json.dump(dict(_meta=dict(what="HIP path vs float64 referee at BASELINE config shapes (tests/scale_cases.py)", lib=os.environ.get("SPARF_LIB", "default")),
               summary=summary, cases=results), open(args.out, "w"), indent=1)
print("wrote", args.out)
