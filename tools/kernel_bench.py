"""Per-kernel timings on the GPU (stream events, single-kernel launches through the C ABI).
Usage: python tools/kernel_bench.py [bf16|fp32] [rays] [nsamp]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]
from sparf_amd import lib as L, ops                       # noqa: E402
from sparf_amd.config import baseline_opt                 # noqa: E402
from sparf_amd.renderer import Graph                      # noqa: E402


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    prec_name = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    rays = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 192
    prec = L.PREC_IDS[prec_name]
    dev = torch.device("cuda:0")
    opt = baseline_opt(1, hip=dict(precision=prec_name))
    torch.manual_seed(0)
    graph = Graph(opt, dev)
    if os.environ.get("SPARF_ZERO_WEIGHTS"):      # DVFS probe: same instruction stream, no operand toggling (MI355X_MICROARCH.md "DVFS give-back")
        with torch.no_grad():
            for p in graph.parameters():
                p.zero_()
    lib = L.load()
    g = torch.Generator().manual_seed(3)
    c = (torch.rand(rays, 3, generator=g) - 0.5 + torch.tensor([0.0, 0.0, -3.0])).to(dev)
    d = (torch.rand(rays, 3, generator=g) * 0.6 - 0.3 + torch.tensor([0.0, 0.0, 1.0])).to(dev)
    t = (torch.sort(torch.rand(rays, N, generator=g), dim=1).values * 4.0 + 1.2).to(dev)
    net = graph.nerf_fine
    s = L.stream_ptr(dev)
    print("pack       %.3f ms" % timeit(lambda: ops.pack_weights(net.hip_params(), prec)))
    packed = net.packed(prec)
    c2f = net.band_weights()
    fa, out, save, k1 = ops.build_pass_fwd(prec, c, d, t, None, 0.0, False, packed, c2f, True)
    fa0, out0, _, k0 = ops.build_pass_fwd(prec, c, d, t, None, 0.0, False, packed, c2f, False)
    L.check(lib.sparf_pass_forward(ctypes.byref(fa), s), "fwd")
    grads = (torch.rand(rays, 3, device=dev), None, None, None)
    ba, gp, _, _, k2 = ops.build_pass_bwd(prec, c, d, t, None, 0.0, False, packed, c2f, save, out, grads, False)
    bap, gpp, dc, dd, k3 = ops.build_pass_bwd(prec, c, d, t, None, 0.0, False, packed, c2f, save, out, grads, True)
    L.check(lib.sparf_pass_backward(ctypes.byref(ba), s), "bwd")
    L.check(lib.sparf_pass_backward(ctypes.byref(bap), s), "bwd pose")
    rows = rays * N
    fl = rows * 2 * 527872 / 1e9
    K = lambda which, a, b: (lambda: L.check(lib.sparf_launch_kernel(which, ctypes.byref(a), ctypes.byref(b), s), "k"))
    kernels = [("fwd save", K(0, fa, ba)), ("fwd nosave", K(0, fa0, ba)), ("dgrad", K(1, fa, ba)), ("dgrad pose", K(1, fa, bap)), ("wgrad", K(2, fa, ba))]
    if prec_name == "bf16x3" and not os.environ.get("SPARF_ABI_ANY"):      # both workgroup geometries of the bf16x3 data-gradient kernel, pinned (sparf_hip.h which = 3 / 4)
        kernels += [("dgrad 8w", K(3, fa, ba)), ("dgrad 4w", K(4, fa, ba)), ("dgrad pose 8w", K(3, fa, bap)), ("dgrad pose 4w", K(4, fa, bap))]
    if os.environ.get("KB_ONLY"):
        only = os.environ["KB_ONLY"].split(",")           # name prefixes; "name$" = exactly that name
        kernels = [k for k in kernels if any((k[0] == o[:-1]) if o.endswith("$") else k[0].startswith(o) for o in only)]
    for name, fn in kernels:
        ms = timeit(fn)
        print(f"{name:11s}{ms:8.3f} ms   {fl / ms:8.1f} TFLOP/s-equiv   rows {rows}")
        prof_fn = "sparf_debug_prof" if name.startswith("fwd") else "sparf_debug_prof_bwd" if name.startswith("dgrad") else None
        if prof_fn and hasattr(lib, prof_fn):      # SP_PROF builds: wave-time accounting (mlp_dev.h Prof; the plane-area kernels)
            buf = (ctypes.c_uint64 * 10)()
            torch.cuda.synchronize()
            getattr(lib, prof_fn)(buf)
            tot = sum(buf) or 1
            names = ("barrier", "dma_issue", "lds+mfma", "epilogue", "stores", "tile end", "staged inputs", "encoding", "x0 build", "after layer 9")
            if prof_fn.endswith("_bwd"):        # mlp_bwd_impl.h g_prof_bwd
                names = ("barrier", "dma_issue", "lds+mfma", "exposed epilogue", "mask loads + stores + acc clear", "tile end (pose: encoding backward)", "tile inputs", "-", "-", "-")
            print("    wave 0 cycles: " + ", ".join(f"{n} {v / tot * 100:.1f}%" for n, v in zip(names, buf)) + f"  (total {tot / 1e6:.2f} M)")
    if os.environ.get("KB_ONLY") and "pass" not in os.environ["KB_ONLY"]:
        return
    print("pass fwd   %.3f ms" % timeit(lambda: L.check(lib.sparf_pass_forward(ctypes.byref(fa), s), "f")))
    print("pass bwd   %.3f ms" % timeit(lambda: L.check(lib.sparf_pass_backward(ctypes.byref(ba), s), "b")))


if __name__ == "__main__":
    main()
