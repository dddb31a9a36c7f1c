"""Host-side cost of one Graph.render + backward at a tiny batch.  Two numbers per mode:
  host  : the time the calling thread needs to ISSUE one call (forward: until render() returns; + backward: until loss.backward()
          returns, i.e. until the autograd engine has enqueued everything), measured with the GPU idle at the start of every call
          (torch.cuda.synchronize() before the clock starts) -- the package's Python / ctypes / allocator path and nothing else;
  wall  : calls issued back to back, one synchronize at the end: max(host, GPU) -- at 64 rays the ~25 kernels of a render +
          backward have a latency floor of their own (one 128-row tile through ten layers), so this is NOT host time any more
          once the host path is short (round 4 quoted it as such).
Usage: python tests/tools/host_overhead.py [precision] [--profile]"""
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


TIMERS = {}


def instrument():
    """time spent inside this package's autograd Functions (their backward runs on the autograd engine's thread, which cProfile of
    the calling thread does not see): the rest of `loss.backward()` is the engine itself -- AccumulateGrad of 40 parameters, the
    loss's own backward kernels"""
    from sparf_amd import ops
    for cls in (ops.RenderFn, ops.RayGen, ops.NerfPass):
        for name in ("forward", "backward"):
            fn = getattr(cls, name)

            def timed(*a, __fn=fn, __key=f"{cls.__name__}.{name}", **k):
                t0 = time.perf_counter()
                try:
                    return __fn(*a, **k)
                finally:
                    e = TIMERS.setdefault(__key, [0.0, 0])
                    e[0] += time.perf_counter() - t0
                    e[1] += 1
            setattr(cls, name, staticmethod(timed))


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "bf16x3"
    dev = torch.device("cuda:0")
    instrument()
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

            def one():
                if bw is None:
                    with torch.no_grad():
                        step(False)
                else:
                    step(bw)
            t0 = time.perf_counter()
            for _ in range(n):
                one()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / n * 1e6
            host = 0.0
            for _ in range(n):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                one()
                host += time.perf_counter() - t1
            torch.cuda.synchronize()
            print(f"config {cfg} [{prec}] {name}: host {host / n * 1e6:.0f} us, wall {wall:.0f} us per render call (64 rays x (64+128) samples)")
            if bw:
                print("    inside the package's autograd Functions, us per call: " + ", ".join(f"{k} {v[0] / v[1] * 1e6:.0f}" for k, v in sorted(TIMERS.items()) if v[1]))
            TIMERS.clear()
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
