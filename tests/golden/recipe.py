"""Deterministic inputs shared by the golden generator (runs the REFERENCE) and
the tests (run the oracle / the HIP path).  numpy's legacy RandomState stream
is stable across numpy versions, so both sides rebuild identical weights,
cameras and rays from a seed instead of storing megabytes of weights."""
import math

import numpy as np
import torch

from sparf_amd.config import default_opt


def small_opt(**over):
    """Shipped architecture (8x256 + skip@4, 128-wide colour branch) with few
    samples so fixtures stay small."""
    # (hip.precision: the tests built on this option tree hold the HIP path to the reference's golden vectors at fp32-level
    # bounds, so they pin the exact-fp32-MFMA mode; the product default, bf16x3, is the mode of every test parametrized over
    # `precision` and of tests/test_api_cpu.py::test_default_precision_is_the_headline_mode.  The reference ignores the key.)
    base = dict(nerf=dict(sample_intvs=8, sample_intvs_fine=8, fine_sampling=True, rand_rays=16,
                          depth=dict(param="metric", range=[1, 0])), hip=dict(precision="fp32"))
    o = default_opt(**base)
    from sparf_amd.config import _merge
    _merge(o, over)
    return o


def layer_list(opt):
    d3 = 3 + 6 * opt.arch.posenc.L_3D
    dv = 3 + 6 * opt.arch.posenc.L_view
    lf = opt.arch.layers_feat
    out = []
    n = len(lf) - 1
    for li in range(n):
        k_in = d3 if li == 0 else lf[li]
        if li in opt.arch.skip:
            k_in += d3
        k_out = lf[li + 1] + (1 if li == n - 1 else 0)
        out.append((f"mlp_feat.{li}", k_out, k_in))
    lr = opt.arch.layers_rgb
    for li in range(len(lr) - 1):
        k_in = lf[-1] + dv if li == 0 else lr[li]
        out.append((f"mlp_rgb.{li}", lr[li + 1], k_in))
    return out


def make_state_dict(opt, seed, progress=None):
    """Xavier-scaled uniform weights and small non-zero biases (the reference
    zero-inits biases; non-zero ones make the bias path observable)."""
    rs = np.random.RandomState(seed)
    sd = {}
    for name, k_out, k_in in layer_list(opt):
        a = math.sqrt(2.0) * math.sqrt(6.0 / (k_in + k_out))
        sd[name + ".weight"] = torch.from_numpy(rs.uniform(-a, a, size=(k_out, k_in)).astype(np.float32))
        sd[name + ".bias"] = torch.from_numpy(rs.uniform(-0.05, 0.05, size=(k_out,)).astype(np.float32))
    if progress is None:
        progress = 1.0 if opt.barf_c2f is None else 0.0
    sd["progress"] = torch.tensor(float(progress))
    return sd


def ring_cameras(B, seed=0, radius=3.2, H=6, W=8, f=7.0):
    """B world-to-camera poses [B,3,4] on a ring looking at the origin (+z
    forward, as the reference assumes) and intrinsics [B,3,3]."""
    rs = np.random.RandomState(1000 + seed)
    poses = []
    for b in range(B):
        ang = 2 * math.pi * b / max(B, 1) + rs.uniform(-0.1, 0.1)
        elev = rs.uniform(-0.2, 0.3)
        c = np.array([radius * math.cos(ang) * math.cos(elev), radius * math.sin(elev),
                      radius * math.sin(ang) * math.cos(elev)])
        z = -c / np.linalg.norm(c)
        up = np.array([0.0, 1.0, 0.0])
        x = np.cross(up, z); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R_c2w = np.stack([x, y, z], axis=1)
        R_w2c = R_c2w.T
        t = -R_w2c @ c
        poses.append(np.concatenate([R_w2c, t[:, None]], axis=1))
    pose = torch.from_numpy(np.stack(poses).astype(np.float32))
    K = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], dtype=torch.float32)
    return pose, K[None].repeat(B, 1, 1)
