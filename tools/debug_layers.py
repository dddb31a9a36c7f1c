"""GPU diagnostic (not a test): run one forward pass with activation saving and compare
every saved buffer, row by row, with the CPU emulation of tests/test_layout_emulation.py.
Localises a wrong layer / permutation / hardware-layout assumption in one GPU run.
Usage: python tools/debug_layers.py [bf16|fp32]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]

from oracle import nerf_oracle as O                       # noqa: E402
from sparf_amd import lib as L, ops                         # noqa: E402
from tests.golden.recipe import small_opt, make_state_dict  # noqa: E402
from tests.test_layout_emulation import Emu, flat_params    # noqa: E402
import ctypes                                               # noqa: E402


def main(prec_name):
    prec = L.PREC_IDS[prec_name]
    dev = torch.device("cuda:0")
    opt = small_opt()
    sd = make_state_dict(opt, 5)
    R, N = 5, 8
    rs = np.random.RandomState(0)
    center = torch.from_numpy(rs.uniform(-0.5, 0.5, size=(R, 3)).astype(np.float32)) + torch.tensor([0.0, 0.0, -3.0])
    dirs = torch.from_numpy(rs.uniform(-0.3, 0.3, size=(R, 3)).astype(np.float32)) + torch.tensor([0.0, 0.0, 1.0])
    t = torch.from_numpy(np.sort(rs.uniform(1.2, 5.2, size=(R, N)), axis=1).astype(np.float32))
    plist = [sd[f"{n}.{k}"].to(dev) for n in L.PARAM_NAMES for k in ("weight", "bias")]
    packed = ops.pack_weights(plist, sd["progress"].to(dev), None, prec)
    lib = L.load()
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    out = dict(raylen=f(R), sigma_raw=f(R, N), rgb_samples=f(R, N, 3), density=f(R, N), weights=f(R, N), rgb=f(R, 3),
               depth=f(R), opacity=f(R), depth_var=f(R), rgb_var=f(R), all_cumulated=f(R))
    rows = R * N
    save = torch.zeros(lib.sparf_save_bytes(prec, rows), dtype=torch.uint8, device=dev)
    venc = torch.empty(R * 32 * 4, dtype=torch.uint8, device=dev)
    c, d, tt = center.to(dev), dirs.to(dev), t.to(dev)
    a = L.PassFwd(prec=prec, nrays=R, nsamp=N, center=c.data_ptr(), dir=d.data_ptr(), t=tt.data_ptr(), noise=None, noise_scale=0.0,
                  white_bg=0, packed=packed.data_ptr(), save=save.data_ptr(), venc_ws=venc.data_ptr(),
                  **{k: v.data_ptr() for k, v in out.items()})
    L.check(lib.sparf_pass_forward(ctypes.byref(a), L.stream_ptr(dev)), "fwd")
    torch.cuda.synchronize()
    adt = torch.bfloat16 if prec == L.PREC_BF16 else torch.float32
    rows_pad = (rows + 31) // 32 * 32
    sv = save.view(adt)[: rows_pad * 2272].float().cpu().numpy()
    cols = [("XS", 320), ("H0", 256), ("H1", 256), ("H2", 256), ("H4", 256), ("H5", 256), ("H6", 256), ("FV", 288), ("G", 128)]
    CH = 8 if prec == L.PREC_BF16 else 4
    bufs, o = {}, 0
    for name, cnum in cols:
        # tile-major: [tile32][chunk][row&31][CH] -> [rows][cols]
        tiles = sv[o:o + rows_pad * cnum].reshape(rows_pad // 32, cnum // CH, 32, CH)
        bufs[name] = tiles.transpose(0, 2, 1, 3).reshape(rows_pad, cnum)[:rows]
        o += rows_pad * cnum
    emu = Emu(prec, flat_params(sd))
    P = emu.to_pos
    pts = O.points_from_depth(center[None], dirs[None], t[None, :, :, None]).double()
    x0 = torch.cat([pts, O.positional_encoding(opt, pts, 10, sd["progress"])], -1)[0].reshape(rows, 63).numpy()
    dn = torch.nn.functional.normalize(dirs.double(), dim=-1)
    vv = torch.cat([dn, O.positional_encoding(opt, dn, 4, sd["progress"])], -1).numpy()
    worst = {}
    for row in range(rows):
        fw = emu.forward(x0[row], vv[row // N])
        ac = fw["acts"]
        exp = {"XS": np.concatenate([P(ac["out3"]), P(ac["x0"])]), "H0": P(ac["out0"]), "H1": P(ac["out1"]), "H2": P(ac["out2"]),
               "H4": P(ac["out4"]), "H5": P(ac["out5"]), "H6": P(ac["out6"]), "FV": np.concatenate([P(ac["out7"]), P(ac["v"])]),
               "G": P(ac["out8"])}
        for k, e in exp.items():
            err = np.abs(bufs[k][row] - e).max() / (np.abs(e).max() + 1e-30)
            worst[k] = max(worst.get(k, 0), err)
        worst["sigma_raw"] = max(worst.get("sigma_raw", 0), abs(out["sigma_raw"].view(-1)[row].item() - fw["sigma_raw"]))
        zz = 1 / (1 + np.exp(-fw["z"]))
        worst["rgb"] = max(worst.get("rgb", 0), np.abs(out["rgb_samples"].view(-1, 3)[row].cpu().numpy() - zz).max())
    print(prec_name, "max relative error per saved buffer (order of computation: XS.x0 H0 H1 H2 XS.h3 H4 H5 H6 FV G):")
    for k, v in worst.items():
        print(f"  {k:10s} {v:.3e}")
    # x0 part separately (first thing computed)
    print("  x0 cols err:", np.abs(bufs["XS"][:, 256:] - np.stack([P(emu.forward(x0[r], vv[r // N])['acts']['x0']) for r in range(rows)])).max())


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "fp32")
