"""CPU speed of the REFERENCE modules themselves vs the oracle ("port") on identical inputs.

Runs only in the build container (needs /root/reference; the GPU box has no reference tree, so
bench.py's `cpu_baseline` leg times the port there).  This script pins that leg to the reference:
both are timed here on the same host, same thread count, same rays / weights, BASELINE
config 0 (3 views x 85 rays = 255 rays; 64 coarse + 128 fine, and 64 coarse only), forward +
backward, and the ratio port / reference is committed as profiles/r03_cpu_ref_vs_port.json.
bench.py quotes it as `cpu_baseline.port_over_reference`.

    python tests/tools/cpu_ref_vs_port.py [--out profiles/r03_cpu_ref_vs_port.json]

Reference entry point timed: source/models/renderer.py:250-345 (Graph.render) + autograd backward.
"""
import argparse
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat"), "/root/reference"]

from source.models.renderer import Graph as RefGraph              # noqa: E402  (the reference)
from oracle import nerf_oracle as O                                # noqa: E402
from bench_workloads import cameras                                # noqa: E402
from sparf_amd.config import baseline_opt                          # noqa: E402


def cpu_model():
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            return line.split(":", 1)[1].strip()
    return "unknown"


def median_time(fn, n=7, budget=60.0):
    fn()
    ts, t0 = [], time.perf_counter()
    while len(ts) < n and (time.perf_counter() - t0 < budget or len(ts) < 3):
        ts.append(fn())
    ts.sort()
    return ts[len(ts) // 2], len(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_cpu_ref_vs_port.json"))
    ap.add_argument("--threads", type=int, nargs="*", default=None)
    args = ap.parse_args()
    ncpu = os.cpu_count() or 1
    threads = args.threads or sorted({1, min(4, ncpu), ncpu})
    B, R, H, W = 3, 85, 300, 400
    pose, intr = cameras(2, "cpu")
    g = torch.Generator().manual_seed(1)
    idx = torch.randperm(H * W, generator=g)[:R]
    rng = torch.tensor([1.2, 5.2])
    results = []
    for fine in (True, False):
        opt = baseline_opt(0)
        opt.nerf.fine_sampling = fine
        torch.manual_seed(0)
        ref = RefGraph(opt, torch.device("cpu"))
        pc = {k: v.detach().clone().requires_grad_(k != "progress") for k, v in ref.nerf.state_dict().items()}
        pf = {k: v.detach().clone().requires_grad_(k != "progress") for k, v in ref.nerf_fine.state_dict().items()} if fine else None
        Nc, Nf = opt.nerf.sample_intvs, opt.nerf.sample_intvs_fine
        center, ray = O.rays_at_index(pose, intr, H, W, idx)

        def ref_iter():
            t0 = time.perf_counter()
            out = ref.render(opt, pose, H=H, W=W, intr=intr, ray_idx=idx, depth_range=rng, iter=1000, mode="train")
            loss = out.rgb.mean() + (out.rgb_fine.mean() if fine else 0.0)
            loss.backward()
            dt = time.perf_counter() - t0
            ref.zero_grad(set_to_none=True)
            return dt

        def port_iter():
            jitter, grid = torch.rand(B, R, Nc, 1, generator=g), torch.rand(Nf + 1, generator=g)
            nc, nf = torch.randn(B, R, Nc, generator=g), torch.randn(B, R, Nc + Nf, generator=g)
            t0 = time.perf_counter()
            out = O.render(opt, pc, pf, center, ray, [rng[0], rng[1]], mode="train", it=1000, jitter=jitter, grid=grid, noise_c=nc, noise_f=nf)
            (out["rgb"].mean() + (out["rgb_fine"].mean() if fine else 0.0)).backward()
            dt = time.perf_counter() - t0
            for p in (pc, pf):
                if p is not None:
                    for v in p.values():
                        v.grad = None
            return dt

        for n in threads:
            torch.set_num_threads(n)
            tr, nr = median_time(ref_iter)
            tp, npt = median_time(port_iter)
            e = dict(samples="64+128" if fine else "64 coarse only", threads=n, rays=B * R,
                     reference_rays_per_s=B * R / tr, port_rays_per_s=B * R / tp, port_over_reference=tr and (B * R / tp) / (B * R / tr),
                     reference_ms=tr * 1e3, port_ms=tp * 1e3, iterations=(nr, npt))
            print(json.dumps(e), flush=True)
            results.append(e)
    full = [e for e in results if e["samples"] == "64+128"]
    best = max(full, key=lambda e: e["reference_rays_per_s"])
    summary = dict(port_over_reference=best["port_over_reference"], threads=best["threads"],
                   reference_rays_per_s=best["reference_rays_per_s"], port_rays_per_s=best["port_rays_per_s"],
                   port_over_reference_range=[min(e["port_over_reference"] for e in full), max(e["port_over_reference"] for e in full)])
    doc = dict(what="reference Graph.render + backward (source/models/renderer.py:250-345 under torch autograd) vs oracle/nerf_oracle.py "
                    "render + backward, same host, same threads, same rays / weights; BASELINE config 0 (255 rays)",
               host=dict(cpu=cpu_model(), threads_available=ncpu, torch=torch.__version__), summary=summary, runs=results)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote", args.out, json.dumps(summary))


if __name__ == "__main__":
    main()
