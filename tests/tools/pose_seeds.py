"""Does the default mode TRAIN like the reference, as a distribution (second half): tests/tools/psnr_curve.py -- the
fp32 oracle (PyTorch-ROCm ops + torch autograd + torch.optim.Adam), HIP fp32 and HIP bf16x3 trained side by side from identical
initialisation on identical rays and draws -- repeated over seeds (initial weights, initial pose noise, ray / draw stream), final
held-out PSNR and pose error (degrees) as mean +- std per trainer and as the paired difference to the oracle.

    python tests/tools/pose_seeds.py --config 2 --seeds 5 --steps 1000 --out gpurun_out/r05_pose_seeds_c2.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]

import numpy as np  # noqa: E402

from tests.tools import psnr_curve as PC  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--seeds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--modes", default="bf16x3,fp32")
    ap.add_argument("--max-seconds", type=float, default=600.0, help="per seed")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    runs = []
    for seed in range(a.seeds):
        args = PC.parse(["--config", str(a.config), "--steps", str(a.steps), "--modes", a.modes, "--eval-every", str(a.steps), "--eval-rays", "4096",
                         "--grad-check-at", "-1", "--max-seconds", str(a.max_seconds), "--seed", str(seed), "--quiet"])
        doc = PC.run(args)
        fin = {k: v for k, v in doc["final"].items() if isinstance(v, dict)}
        start = {k: v for k, v in doc["curve"][0].items() if isinstance(v, dict)}
        runs.append(dict(seed=seed, steps=doc["steps_done"], seconds=doc["seconds"], final=fin, start=start))
        print(json.dumps(runs[-1]), flush=True)
    names = sorted(runs[0]["final"])
    summary = {}
    for n in names:
        for key in ("psnr", "pose_err_deg"):
            vals = [r["final"][n][key] for r in runs if key in r["final"][n]]
            if vals:
                summary.setdefault(n, {})[key] = dict(mean=float(np.mean(vals)), std=float(np.std(vals, ddof=1)) if len(vals) > 1 else 0.0, values=vals)
            if n != "oracle_fp32" and "oracle_fp32" in runs[0]["final"]:
                d = [r["final"][n][key] - r["final"]["oracle_fp32"][key] for r in runs if key in r["final"][n]]
                if d:
                    summary[n][key + "_minus_oracle"] = dict(mean=float(np.mean(d)), std=float(np.std(d, ddof=1)) if len(d) > 1 else 0.0, values=d)
    doc = dict(what=__doc__.split("\n\n")[0], config=a.config, steps=a.steps, seeds=a.seeds, summary=summary, runs=runs)
    print(json.dumps(summary, indent=1))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(doc, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
