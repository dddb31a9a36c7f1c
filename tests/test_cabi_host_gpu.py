"""The drop-in boundary without Python in the loop: tools/cabi_host_example.cpp is a plain C++ host that drives one
network pass (forward + backward) through include/sparf_hip.h -- hipMalloc'd buffers, `extern "C"` calls, no torch
types.  The test builds it (hipcc, linked against the in-tree libsparf_hip.so), feeds it seeded inputs through a
file and checks outputs and gradients against the oracle.  Run with `pytest -m gpu`."""
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from sparf_amd import lib as L
from tests.golden.recipe import make_state_dict, small_opt

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _binary(name="cabi_host_example"):
    src = os.path.join(ROOT, "tools", name + ".cpp")
    exe = os.path.join(ROOT, "tools", name + ".out")
    so = os.path.join(ROOT, "sparf_amd", "libsparf_hip.so")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(so)):
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-std=c++17", "-O2", src, "-I" + os.path.join(ROOT, "include"),
                               "-L" + os.path.join(ROOT, "sparf_amd"), "-lsparf_hip", "-Wl,-rpath," + os.path.join(ROOT, "sparf_amd"), "-o", exe])
    return exe


@pytest.mark.parametrize("prec_name,tol,gtol", [("fp32", 1e-4, 5e-3), ("bf16x3", 1e-4, 5e-2)])
def test_cpp_host_over_the_c_abi(tmp_path, prec_name, tol, gtol):
    R, N = 150, 40                       # 6000 rows: ragged against every tile size
    opt = small_opt(nerf=dict(setbg_opaque=True))
    sd = make_state_dict(opt, 77)
    rs = np.random.RandomState(3)
    center = (rs.uniform(-0.5, 0.5, size=(R, 3)) + [0, 0, -3.0]).astype(np.float32)
    dirs = (rs.uniform(-0.3, 0.3, size=(R, 3)) + [0, 0, 1.0]).astype(np.float32)
    t = np.sort(rs.uniform(1.2, 5.2, size=(R, N)), axis=1).astype(np.float32)
    g_rgb = rs.uniform(-1, 1, size=(R, 3)).astype(np.float32)
    g_depth = rs.uniform(-1, 1, size=(R,)).astype(np.float32)
    names = [f"{n}.{k}" for n in L.PARAM_NAMES for k in ("weight", "bias")]
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        np.array([L.PREC_IDS[prec_name], R, N, 1], dtype=np.int32).tofile(f)
        for k in names:
            sd[k].numpy().astype(np.float32).tofile(f)
        for a in (center, dirs, t, g_rgb, g_depth):
            a.tofile(f)
    r = subprocess.run([_binary(), str(fin), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    # the ray-segment table driven from C++ (two render calls in one pass; the host compares the backwards itself and exits 3 on a mismatch)
    assert "segments: both active 0.00e+00" in r.stdout, r.stdout
    out = np.fromfile(fout, dtype=np.float32)
    sizes = [R * 3, R, R, R * N, L.N_PARAMS, R * 3, R * 3]
    assert out.size == sum(sizes)
    rgb, depth, opacity, weights, gparams, d_center, d_dir = np.split(out, np.cumsum(sizes)[:-1])
    # oracle on the same inputs
    sdo = {k: v.clone().requires_grad_(k != "progress") for k, v in sd.items()}
    c, d = torch.from_numpy(center)[None].requires_grad_(True), torch.from_numpy(dirs)[None].requires_grad_(True)
    ref = O.pass_fixed(opt, sdo, c, d, torch.from_numpy(t)[None, :, :, None], mode="val")
    ((ref["rgb"][0] * torch.from_numpy(g_rgb)).sum() + (ref["depth"][0, :, 0] * torch.from_numpy(g_depth)).sum()).backward()
    rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
    assert rel(rgb, ref["rgb"].detach().numpy().ravel()) < tol and rel(depth, ref["depth"].detach().numpy().ravel()) < tol
    assert rel(opacity, ref["opacity"].detach().numpy().ravel()) < tol and rel(weights, ref["weights"].detach().numpy().ravel()) < tol
    gref = np.concatenate([sdo[k].grad.numpy().ravel() for k in names])
    assert float(np.linalg.norm(gparams - gref) / np.linalg.norm(gref)) < gtol
    assert rel(d_center, c.grad.numpy().ravel()) < gtol * 2 and rel(d_dir, d.grad.numpy().ravel()) < gtol * 2


@pytest.mark.parametrize("prec_name", ["fp32", "bf16x3"])
def test_cpp_training_loop_over_the_c_abi(tmp_path, prec_name):
    """tools/cabi_train_example.cpp: the whole hierarchical render + backward + clip + Adam iteration as extern "C"
    calls from a C++ host.  Fed the same initial weights, cameras, targets and per-step draws as the Python `Graph`
    path (fused loss, FusedAdam): the first loss agrees to 1e-6 (same kernels, same inputs), the curves stay together
    and both train."""
    from sparf_amd import ops
    from sparf_amd.optim import FusedAdam
    from sparf_amd.renderer import Graph
    from tests.golden.recipe import ring_cameras
    from tests.scale_cases import injected_rng
    dev = torch.device("cuda:0")
    B, H, W, R, Nc, Nf, steps = 2, 20, 24, 96, 16, 32, 12
    dmin, dmax, lr, clip = np.float32(1.2), np.float32(5.2), 1e-3, 0.1
    opt = small_opt(nerf=dict(sample_intvs=Nc, sample_intvs_fine=Nf, rand_rays=B * R), hip=dict(precision=prec_name))
    sd_c, sd_f = make_state_dict(opt, 301), make_state_dict(opt, 302)
    pose, intr = ring_cameras(B, H=H, W=W, f=20.0)
    rs = np.random.RandomState(9)
    image = rs.uniform(size=(B, H * W, 3)).astype(np.float32)
    names = [f"{n}.{k}" for n in L.PARAM_NAMES for k in ("weight", "bias")]
    draws = []
    fin, fout = tmp_path / "train_in.bin", tmp_path / "train_out.bin"
    with open(fin, "wb") as f:
        np.array([L.PREC_IDS[prec_name], B, W, R, Nc, Nf, steps], dtype=np.int32).tofile(f)
        np.array([dmin, dmax, lr, clip], dtype=np.float32).tofile(f)
        for sd in (sd_c, sd_f):
            for k in names:
                sd[k].numpy().astype(np.float32).tofile(f)
        pose.numpy().astype(np.float32).tofile(f)
        intr.numpy().astype(np.float32).tofile(f)
        for it in range(steps):
            idx = rs.permutation(H * W)[:R].astype(np.int64)
            jitter = rs.uniform(size=(B, R, Nc, 1)).astype(np.float32)
            grid = rs.uniform(size=Nf + 1).astype(np.float32)
            target = image[:, idx]
            idx.tofile(f); jitter.tofile(f); grid.tofile(f); target.tofile(f)
            draws.append((idx, jitter, grid, target))
    r = subprocess.run([_binary("cabi_train_example"), str(fin), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    loss_cpp = np.fromfile(fout, dtype=np.float32)
    assert loss_cpp.shape == (steps,)
    # the same iteration through the Python mirror
    graph = Graph(opt, dev)
    graph.nerf.load_state_dict(sd_c)
    graph.nerf_fine.load_state_dict(sd_f)
    optim = FusedAdam([graph.nerf, graph.nerf_fine], lr=lr, max_grad_norm=clip)
    rng = torch.tensor([dmin, dmax], device=dev)
    loss_py = []
    for idx, jitter, grid, target in draws:
        optim.zero_grad(set_to_none=True)
        with injected_rng(torch.from_numpy(jitter), torch.from_numpy(grid), []):
            ret = graph.render(opt, pose.to(dev), H=H, W=W, intr=intr.to(dev), ray_idx=torch.from_numpy(idx).to(dev), depth_range=rng, iter=1, mode="train")
        loss = ops.photometric_loss(ret.rgb, torch.from_numpy(target).to(dev), rgb_fine=ret.rgb_fine)
        loss.backward()
        optim.step()
        loss_py.append(float(loss.detach()))
    loss_py = np.array(loss_py, dtype=np.float32)
    print(prec_name, "C++ host", loss_cpp[[0, -1]], "python mirror", loss_py[[0, -1]])
    assert abs(loss_cpp[0] - loss_py[0]) <= 1e-6 * loss_py[0]
    assert np.abs(loss_cpp - loss_py).max() <= 2e-2 * loss_py.max()
    assert loss_cpp[-3:].mean() < loss_cpp[:3].mean()
