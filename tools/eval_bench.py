"""Full-image inference throughput: Graph.render_by_slices over B views of
300x400 pixels, 64 coarse + 128 fine samples, deterministic sampling, no gradients.
Usage: python tools/eval_bench.py [bf16|fp32] [views]"""
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
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda:0")
    H, W = 300, 400
    opt = baseline_opt(1, hip=dict(precision=prec))
    torch.manual_seed(0)
    graph = Graph(opt, dev)
    pose, intr = cameras(1, dev)                          # config 1: four views on a ring, 300x400
    pose, intr = pose[:B], intr[:B]
    image = analytic_images(pose, intr, H, W)
    with torch.no_grad():
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ret = graph.render_by_slices(opt, pose, H=H, W=W, intr=intr, depth_range=[1.2, 5.2], iter=None, mode="val")
            mse = ((ret["rgb_fine"].view(B, H, W, 3).permute(0, 3, 1, 2) - image) ** 2).mean()
            psnr = -10 * torch.log10(mse)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"{prec}: {B} x {H}x{W} = {B * H * W} rays x (64+128) in {dt * 1e3:.1f} ms -> {B * H * W / dt / 1e6:.2f} M rays/s "
                  f"({B * H * W * 256 * 2 * 527872 / dt / 1e12:.0f} TFLOP/s-equiv), PSNR vs the analytic target (random-init network) {float(psnr):.2f} dB")


if __name__ == "__main__":
    main()
