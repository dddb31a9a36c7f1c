"""Host-side cost of one Graph.render + backward: tiny batches (the kernels take microseconds), so the wall time per call is this
package's Python / ctypes path.  Usage: python tests/tools/host_overhead.py [precision] [--profile]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]
from bench_workloads import config_opt                   # noqa: E402
from sparf_amd.renderer import Graph                      # noqa: E402
from tests.golden.recipe import ring_cameras              # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "bf16x3"
    dev = torch.device("cuda:0")
    for cfg in (1, 3):
        opt = config_opt(cfg, prec, rays=64)
        torch.manual_seed(0)
        graph = Graph(opt, dev)
        H, W, B = 30, 40, 2
        pose, intr = ring_cameras(B, H=H, W=W)
        pose, intr = pose.to(dev), intr.to(dev)
        idx = torch.arange(32, device=dev)

        def step(backward=True):
            ret = graph.render(opt, pose, H=H, W=W, intr=intr, ray_idx=idx, depth_range=[1.2, 5.2] if cfg == 1 else [1, 0], iter=100, mode="train")
            if backward:
                (ret["rgb"].sum() + ret["rgb_fine"].sum()).backward()

        for _ in range(20):
            step()
        torch.cuda.synchronize()
        for name, bw in (("forward only (no_grad)", None), ("forward", False), ("forward + backward", True)):
            n = 200
            t0 = time.perf_counter()
            if bw is None:
                with torch.no_grad():
                    for _ in range(n):
                        step(False)
            else:
                for _ in range(n):
                    step(bw)
            torch.cuda.synchronize()
            print(f"config {cfg} [{prec}] {name}: {(time.perf_counter() - t0) / n * 1e6:.0f} us per render call (64 rays x (64+128) samples)")
        if "--profile" in sys.argv:
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(200):
                step(True)
            torch.cuda.synchronize()
            pr.disable()
            st = pstats.Stats(pr)
            st.sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
