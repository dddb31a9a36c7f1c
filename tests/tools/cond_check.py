"""Is a d_dir mismatch conditioning or a bug?  Compare the HIP fp32 gradient and the oracle's fp32
gradient with the oracle evaluated in float64.  Usage: python tests/tools/cond_check.py R N"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]
from oracle import nerf_oracle as O                                       # noqa: E402
from sparf_amd import lib as L, ops                                       # noqa: E402
from tests.golden.recipe import small_opt, make_state_dict                # noqa: E402
from tests.test_hip_gpu import make_scene, oracle_forward, params_list, rel_l2   # noqa: E402

R, N = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
opt = small_opt(barf_c2f=[0.4, 0.7], nerf=dict(density_noise_reg=True))
sd = make_state_dict(opt, 9, progress=0.62)
center, dirs, jitter, noise = make_scene(R, N, R * 7 + N)
t = O.sample_depth(opt, 1, R, N, [1.2, 5.2], "train", jitter)[0, :, :, 0]
rs = np.random.RandomState(R + N)
lw = {k: torch.from_numpy(rs.uniform(-1, 1, size=s).astype(np.float32))
      for k, s in (("rgb", (R, 3)), ("depth", (R,)), ("opacity", (R,)), ("weights", (R, N)))}
res = {}
for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
    sdo = {k: v.to(dt) for k, v in sd.items()}
    co, do = center.detach().clone().to(dt).requires_grad_(True), dirs.detach().clone().to(dt).requires_grad_(True)
    ref = oracle_forward(opt, sdo, co, do, t.to(dt), noise.to(dt), "train")
    sum((ref[k].reshape(v.shape) * v.to(dt)).sum() for k, v in lw.items()).backward()
    res[name] = (co.grad.double(), do.grad.double())
plist = params_list(sd, dev)
packed = ops.pack_weights(plist, L.PREC_FP32)
c2f = ops.c2f_weights(sd["progress"].to(dev), opt.barf_c2f, dev)
cg, dg = center.to(dev).requires_grad_(True), dirs.to(dev).requires_grad_(True)
got = ops.nerf_pass(cg, dg, t.to(dev), noise[0].to(dev), 1.0, False, L.PREC_FP32, packed, c2f, plist)
sum((got[k] * v.to(dev)).sum() for k, v in lw.items()).backward()
print("d_dir   : hip32 vs f64 %.2e | oracle32 vs f64 %.2e | hip32 vs oracle32 %.2e" %
      (rel_l2(dg.grad, res["f64"][1]), rel_l2(res["f32"][1], res["f64"][1]), rel_l2(dg.grad, res["f32"][1])))
print("d_center: hip32 vs f64 %.2e | oracle32 vs f64 %.2e | hip32 vs oracle32 %.2e" %
      (rel_l2(cg.grad, res["f64"][0]), rel_l2(res["f32"][0], res["f64"][0]), rel_l2(cg.grad, res["f32"][0])))
