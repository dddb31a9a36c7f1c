"""SHA-256 digests of everything one training pass produces (forward outputs, the saved activations and masks, parameter and
ray gradients) for fixed seeded inputs -- run it under two $SPARF_LIB builds to show that a kernel change is bit-neutral.

    python tools/pass_digest.py bf16x3 ;  SPARF_LIB=sparf_amd/libsparf_hip_base.so python tools/pass_digest.py bf16x3
A second argument `nopose` runs the backward without pose gradients (the other instantiation of the data-gradient kernel)."""
import ctypes
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]
from sparf_amd import lib as L, ops                       # noqa: E402
from sparf_amd.config import baseline_opt                 # noqa: E402
from sparf_amd.renderer import Graph                      # noqa: E402


def digest(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


def main():
    prec_name = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
    rays, N = 1000, 192                                    # 192 000 rows: ragged against the 128 / 256-row tiles
    prec = L.PREC_IDS[prec_name]
    pose = "nopose" not in sys.argv[2:]
    dev = torch.device("cuda:0")
    opt = baseline_opt(2, hip=dict(precision=prec_name))   # config 2: c2f bands active
    torch.manual_seed(0)
    graph = Graph(opt, dev)
    graph.nerf_fine.progress.data.fill_(0.55)
    lib = L.load()
    g = torch.Generator().manual_seed(3)
    c = (torch.rand(rays, 3, generator=g) - 0.5 + torch.tensor([0.0, 0.0, -3.0])).to(dev)
    d = (torch.rand(rays, 3, generator=g) * 0.6 - 0.3 + torch.tensor([0.0, 0.0, 1.0])).to(dev)
    t = (torch.sort(torch.rand(rays, N, generator=g), dim=1).values * 4.0 + 1.2).to(dev)
    net = graph.nerf_fine
    s = L.stream_ptr(dev)
    packed, c2f = net.packed(prec), net.band_weights()
    fa, out, save, k1 = ops.build_pass_fwd(prec, c, d, t, None, 0.0, False, packed, c2f, True)
    L.check(lib.sparf_pass_forward(ctypes.byref(fa), s), "fwd")
    grads = (torch.rand(rays, 3, generator=g).to(dev), torch.rand(rays, generator=g).to(dev), None, torch.rand(rays, N, generator=g).to(dev))
    ba, gp, dc, dd, k2 = ops.build_pass_bwd(prec, c, d, t, None, 0.0, False, packed, c2f, save, out, grads, pose)
    L.check(lib.sparf_pass_backward(ctypes.byref(ba), s), "bwd")
    torch.cuda.synchronize()
    items = [(k, v) for k, v in sorted(out.items()) if torch.is_tensor(v)] + [("save_area", save), ("grad_params", gp)] + ([("d_center", dc), ("d_dir", dd)] if pose else [])
    print(prec_name, "pose" if pose else "nopose", os.environ.get("SPARF_LIB", "default"), " ".join(f"{k}:{digest(v)}" for k, v in items))


if __name__ == "__main__":
    main()
