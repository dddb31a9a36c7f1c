"""Debug aid: one coarse pass at BASELINE config 3's inverse-depth samples (t up to ~1e8) -- per-tensor
gradient error of each precision mode against the float64 referee, with the far samples clamped at
several depths and with / without ray gradients.  Usage: python tests/tools/debug_extreme.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]
from oracle import nerf_oracle as O                      # noqa: E402
from sparf_amd import lib as L, ops                       # noqa: E402
from tests import scale_cases as S                       # noqa: E402
from tests.golden.recipe import make_state_dict, ring_cameras   # noqa: E402

dev = torch.device("cuda:0")
cfg = S.CONFIGS[3]
opt = S.case_opt(cfg, "fp32")
R, N = 1024, 64
pose, intr = ring_cameras(1, seed=3, H=cfg["H"], W=cfg["W"], f=cfg["f"])
rs = np.random.RandomState(3)
px = torch.from_numpy(rs.uniform(0, [cfg["W"] - 1, cfg["H"] - 1], size=(1, R, 2)).astype(np.float32))
center, ray = O.rays_at_pixels(pose, intr, px)
jitter = torch.from_numpy(rs.uniform(size=(1, R, N, 1)).astype(np.float32))
t_full = O.sample_depth(opt, 1, R, N, [1, 0], "train", jitter)
sd = make_state_dict(opt, 103, None)
lw = S._loss_weights(rs, {"rgb": (1, R, 3), "depth": (1, R, 1), "opacity": (1, R, 1), "weights": (1, R, N, 1)})
names = [f"{n}.{k}" for n in L.PARAM_NAMES for k in ("weight", "bias")]
lw_all = lw
for keys in (("rgb",), ("depth",), ("opacity",), ("weights",), ("rgb", "depth", "opacity", "weights")):
  lw = {k: lw_all[k] for k in keys}
  for tmax in (1e9, 1e2):
    t = t_full.clamp(max=tmax)
    _, gref, dc, dr = S.referee(opt, sd, sd, center, ray, t, None, None, None, lw, "train", chunk=1024, device="cuda:0")
    _, g32, _, _ = S.referee(opt, sd, sd, center, ray, t, None, None, None, lw, "train", chunk=1024, device="cuda:0", dtype=torch.float32)
    names_ok = [k for k in names if k in gref["nerf"]]
    line = {"ref32": "%.1e" % max(S.rel_l2(g32["nerf"][k], gref["nerf"][k]) for k in names_ok)}
    for prec in ("fp32", "bf16x3"):
        for posegrad in (False,):
            P = L.PREC_IDS[prec]
            plist = [sd[k].to(dev).clone().requires_grad_(True) for k in names]
            packed = ops.pack_weights(plist, P)
            c2f = ops.c2f_weights(sd["progress"].to(dev), None, dev)
            cg, dg = center[0].to(dev).requires_grad_(posegrad), ray[0].to(dev).requires_grad_(posegrad)
            out = ops.nerf_pass(cg, dg, t[0, :, :, 0].to(dev), None, 0.0, False, P, packed, c2f, plist)
            loss = sum((out[k].reshape(lw[k].shape) * lw[k].to(dev)).sum() for k in lw)
            loss.backward()
            errs = {k: S.rel_l2(p.grad, gref["nerf"][k]) for k, p in zip(names, plist) if k in gref["nerf"] and p.grad is not None}
            worst = max(errs, key=errs.get)
            line[prec] = f"{errs[worst]:.1e} ({worst}; L7 {errs['mlp_feat.7.weight']:.1e})"
    print(f"loss {'+'.join(keys)} tmax {tmax:g}:", line, flush=True)
