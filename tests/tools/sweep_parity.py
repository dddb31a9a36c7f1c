"""Size sweep of the pass-level parity (not part of CI): forward + backward of one network pass
against the oracle for row counts around every tiling boundary (wave tile 32, workgroup tiles
128 / 256, wgrad split ranges of 64-row multiples, >1 split) in all precision modes.
Usage: python tests/tools/sweep_parity.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]
from oracle import nerf_oracle as O                                       # noqa: E402
from sparf_amd import lib as L, ops                                       # noqa: E402
from tests.golden.recipe import small_opt, make_state_dict                # noqa: E402
from tests.test_hip_gpu import make_scene, oracle_forward, params_list, rel_err, rel_l2   # noqa: E402

FWD_TOL = {"fp32": 1e-4, "bf16x3": 1e-4, "bf16": 6e-2}
GRAD_TOL = {"fp32": 5e-4, "bf16x3": 3e-2, "bf16": 0.5}       # relative L2 (tiny batches: see DESIGN 2)


def main():
    dev = torch.device("cuda:0")
    opt = small_opt(barf_c2f=[0.4, 0.7], nerf=dict(density_noise_reg=True))
    sd = make_state_dict(opt, 9, progress=0.62)
    worst = {p: [0.0, 0.0] for p in L.PREC_IDS}
    cases = [(1, 2), (1, 31), (2, 16), (3, 11), (31, 1 + 3), (32, 4), (33, 4), (127, 2), (128, 2), (129, 2), (255, 1 + 1), (256, 2),
             (257, 2), (1023, 4), (1024, 4), (1025, 4), (4097, 3), (97, 192)]
    if len(sys.argv) > 1:
        cases = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
    for R, N in cases:
        center, dirs, jitter, noise = make_scene(R, N, R * 7 + N)
        t = O.sample_depth(opt, 1, R, N, [1.2, 5.2], "train", jitter)[0, :, :, 0]
        rs = np.random.RandomState(R + N)
        lw = {k: torch.from_numpy(rs.uniform(-1, 1, size=s).astype(np.float32))
              for k, s in (("rgb", (R, 3)), ("depth", (R,)), ("opacity", (R,)), ("weights", (R, N)))}
        sdo = {k: v.clone().requires_grad_(k != "progress") for k, v in sd.items()}
        co, do = center.clone().requires_grad_(True), dirs.clone().requires_grad_(True)
        ref = oracle_forward(opt, sdo, co, do, t, noise, "train")
        sum((ref[k].reshape(v.shape) * v).sum() for k, v in lw.items()).backward()
        line = f"R={R:5d} N={N:3d} rows={R * N:6d}:"
        for pname, prec in L.PREC_IDS.items():
            plist = [p.clone().requires_grad_(True) for p in params_list(sd, dev)]
            packed = ops.pack_weights(plist, prec)
            c2f = ops.c2f_weights(sd["progress"].to(dev), opt.barf_c2f, dev)
            cg, dg = center.to(dev).requires_grad_(True), dirs.to(dev).requires_grad_(True)
            got = ops.nerf_pass(cg, dg, t.to(dev), noise[0].to(dev), 1.0, False, prec, packed, c2f, plist)
            sum((got[k] * v.to(dev)).sum() for k, v in lw.items()).backward()
            ef = max(rel_err(got[k].reshape(ref[k].shape), ref[k]) for k in ("rgb", "depth", "opacity", "weights", "rgb_samples", "density_samples"))
            names = [f"{n}.{k}" for n in L.PARAM_NAMES for k in ("weight", "bias")]
            errs = {n: rel_l2(p.grad, sdo[n].grad) for p, n in zip(plist, names)}
            errs.update(d_center=rel_l2(cg.grad, co.grad), d_dir=rel_l2(dg.grad, do.grad))
            eg = max(errs.values())
            wname = max(errs, key=errs.get)
            worst[pname] = [max(worst[pname][0], ef), max(worst[pname][1], eg)]
            ok = ef < FWD_TOL[pname] and eg < GRAD_TOL[pname]
            line += f"  {pname} fwd {ef:.1e} grad {eg:.1e}{'' if ok else ' <-- FAIL (' + wname + ')'}"
        print(line, flush=True)
    print("worst:", {k: (f"{v[0]:.1e}", f"{v[1]:.1e}") for k, v in worst.items()})


if __name__ == "__main__":
    main()
