"""SPARF-style iteration: photometric render + 2 correspondence renders +
3 depth-consistency renders (one render_to_max under no_grad), forward + backward, as six
separate calls vs one Graph.render_batch.   Usage: python tools/batch_bench.py [bf16|fp32]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]
from bench_workloads import analytic_images, cameras      # noqa: E402
from sparf_amd.config import baseline_opt                 # noqa: E402
from sparf_amd.renderer import Graph                      # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    dev = torch.device("cuda:0")
    B, H, W = 3, 300, 400
    opt = baseline_opt(1, hip=dict(precision=prec))
    torch.manual_seed(0)
    graph = Graph(opt, dev)
    pose, intr = cameras(1, dev)
    pose, intr = pose[:B], intr[:B]
    image = analytic_images(pose, intr, H, W)
    g = torch.Generator(device=dev).manual_seed(0)

    def requests():
        px = lambda n: torch.rand(n, 2, device=dev, generator=g) * torch.tensor([W, H], device=dev)
        idx = lambda n: torch.randint(0, H * W, (n,), device=dev, generator=g)
        dmax = torch.rand(1, 512, device=dev, generator=g) * 3 + 2
        return [dict(pose=pose, H=H, W=W, intr=intr, ray_idx=idx(682), depth_range=[1.2, 5.2], mode="train"),        # photometric, 2046 rays
                dict(pose=pose[:1], H=H, W=W, intr=intr[:1], pixels=px(512), depth_range=[1.2, 5.2], mode="train"),   # corres, image i
                dict(pose=pose[1:2], H=H, W=W, intr=intr[1:2], pixels=px(512), depth_range=[1.2, 5.2], mode="train"), # corres, image j
                dict(pose=pose[2:], H=H, W=W, intr=intr[2:], pixels=px(512), depth_range=[1.2, 5.2], mode="train"),   # depth-cons, unseen view
                dict(pose=pose[:1], H=H, W=W, intr=intr[:1], pixels=px(512), depth_min=1.2, depth_max=dmax, mode="train", no_grad=True),
                dict(pose=pose[1:2], H=H, W=W, intr=intr[1:2], pixels=px(512), depth_range=[1.2, 5.2], mode="train")]

    def loss_of(rets):
        return sum(r.rgb.mean() + r.rgb_fine.mean() + 0.1 * r.depth_fine.mean() for r in rets if r.rgb.requires_grad)

    def separate():
        rets = []
        for q in requests():
            q = dict(q)
            if q.pop("no_grad", False):
                with torch.no_grad():
                    rets.append(graph.render_to_max(opt, iter=1000, **q))
            else:
                rets.append(graph.render(opt, iter=1000, **q))
        return rets

    for name, fn in (("6 separate calls", separate), ("render_batch", lambda: graph.render_batch(opt, requests(), iter=1000))):
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                graph.zero_grad(set_to_none=True)
                loss_of(fn()).backward()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 10
        nrays = sum((q["pose"].shape[0]) * (len(q.get("ray_idx", q.get("pixels")))) for q in requests())
        print(f"{prec} {name:18s}: {dt * 1e3:7.2f} ms per iteration ({nrays} rays) -> {nrays / dt / 1e3:.0f} k rays/s")


if __name__ == "__main__":
    main()
