"""Where an 8-bit save / gradient area differs from the quantised plane area (development aid of tests/test_q8_saves_gpu.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]
from sparf_amd import lib as L                                                             # noqa: E402
from tests import test_q8_saves_gpu as T                                                   # noqa: E402


def report(got, want, bufs, what):
    bad = (got != want).nonzero()
    print(f"{what}: {bad.shape[0]} of {got.numel()} differ")
    if not bad.shape[0]:
        return
    r, c = bad[:, 0], bad[:, 1]
    edges = torch.tensor([0] + list(torch.tensor(bufs).cumsum(0)), device=c.device)
    b = torch.bucketize(c, edges[1:], right=True)
    lc = c - edges[b]
    half = torch.tensor(bufs, device=c.device)[b] // 2
    h, q = lc // half, lc % half
    for name, v in (("tile", r // 32), ("tile%8", (r // 32) % 8), ("row&31", r % 32), ("buffer", b), ("h", h), ("q%16", q % 16), ("q//16", q // 16)):
        u, n = torch.unique(v, return_counts=True)
        print(f"  by {name}: " + ", ".join(f"{int(a)}:{int(k)}" for a, k in list(zip(u.tolist(), n.tolist()))[:40]))


def main():
    base = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    R, N, pose = 333, 64, True
    opt, sd, center, dirs, t, g = T._inputs(R, N, 5)
    rows = R * N
    o0, s0, w0, *_ = T._pass(L.PREC_IDS[base], opt, sd, center, dirs, t, g, pose)
    o1, s1, w1, *_ = T._pass(L.PREC_IDS[base + "+q8"], opt, sd, center, dirs, t, g, pose)
    X, M = T.decode_planes(s0, T.SAVE_BUFS, 9)
    U, S, Mq = T.decode_q8(s1, T.SAVE_BUFS, 9)
    Ue, Se = T.quantise(X, T.SAVE_BUFS)
    report(U[:rows], Ue[:rows], T.SAVE_BUFS, "save area")
    print("  steps equal:", bool(torch.equal(S[:rows], Se[:rows])), "mismatching step entries:", int((S[:rows] != Se[:rows]).sum()))
    nt = (rows + 255) // 256 * 8
    G, _ = T.decode_planes(w0[:nt * sum(T.GRAD_BUFS) * 64], T.GRAD_BUFS, 0)
    Ug, Sg, _ = T.decode_q8(w1[:nt * (sum(T.GRAD_BUFS) * 32 + len(T.GRAD_BUFS) * 256)], T.GRAD_BUFS, 0)
    Uge, Sge = T.quantise(G, T.GRAD_BUFS)
    report(Ug[:rows], Uge[:rows], T.GRAD_BUFS, "gradient area")


if __name__ == "__main__":
    main()
