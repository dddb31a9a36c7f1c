"""Parity tests proper: the HIP path (through the C ABI) vs the CPU oracle on identical
seeded inputs.  Needs an MI355X: run with `pytest -m gpu`.

Tolerances.  fp32 mode is the parity mode: 1e-4 relative (north_star) on every output and
gradient, measured as max|a-b| / max|b| per tensor (outputs are O(1); a per-element
relative bound is meaningless next to exact zeros).  bf16 mode feeds bf16 operands to the
MFMA (8-bit mantissa) and is checked against a documented, much looser bound.
"""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from sparf_amd import lib as L
from sparf_amd import ops
from tests.golden.recipe import small_opt, make_state_dict

pytestmark = pytest.mark.gpu

# bf16x3 (head + tail bf16 operands, three MFMAs per product) is held to the fp32 bar
TOL = {L.PREC_FP32: 1e-4, L.PREC_BF16: 4e-2, L.PREC_X3: 1e-4}
# gradients: fp32 mode max-norm relative; bf16 mode relative L2 (with ~650 sample rows a
# handful of ReLU masks flipped by bf16 rounding dominate the max norm of a weight gradient)
GTOL = {L.PREC_FP32: 2e-4, L.PREC_BF16: 2.5e-1, L.PREC_X3: 1e-2}


def dev():
    return torch.device("cuda:0")


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def params_list(sd, device):
    return [sd[f"{n}.{k}"].to(device) for n in L.PARAM_NAMES for k in ("weight", "bias")]


def make_scene(R, N, seed, metric=True):
    rs = np.random.RandomState(seed)
    center = torch.from_numpy(rs.uniform(-0.5, 0.5, size=(R, 3)).astype(np.float32)) + torch.tensor([0.0, 0.0, -3.0])
    dirs = torch.from_numpy(rs.uniform(-0.3, 0.3, size=(R, 3)).astype(np.float32)) + torch.tensor([0.0, 0.0, 1.0])
    jitter = torch.from_numpy(rs.uniform(0, 1, size=(1, R, N, 1)).astype(np.float32))
    noise = torch.from_numpy(rs.normal(size=(1, R, N)).astype(np.float32))
    return center, dirs, jitter, noise


def test_sample_coarse_bit_exact():
    R, N = 37, 24
    center, dirs, jitter, _ = make_scene(R, N, 0)
    for param, rng in (("metric", [1.2, 5.2]), ("inverse", [1, 0])):
        opt = small_opt(nerf=dict(depth=dict(param=param)))
        ref = O.sample_depth(opt, 1, R, N, rng, "train", jitter)[0, :, :, 0]
        scale = np.float32(rng[1] - rng[0])
        got = ops.sample_coarse(R, N, rng[0], scale, param == "inverse", dev(), jitter=jitter.to(dev()))
        assert torch.equal(got.cpu(), ref), (param, (got.cpu() - ref).abs().max())
        ref = O.sample_depth(opt, 1, R, N, rng, "val", None)[0, :, :, 0]
        got = ops.sample_coarse(R, N, rng[0], scale, param == "inverse", dev(), u_const=0.5)
        assert torch.equal(got.cpu(), ref)
    dmax = torch.linspace(2.0, 5.0, R)[None]
    ref = O.sample_depth_to_max(N, 1.2, dmax)[0, :, :, 0]
    got = ops.sample_coarse(R, N, 1.2, 0.0, False, dev(), u_const=1.0, dmax_ray=dmax[0].to(dev()))
    assert torch.equal(got.cpu(), ref)


def run_forward(prec, opt, sd, center, dirs, t, noise, mode):
    d = dev()
    plist = params_list(sd, d)
    packed = ops.pack_weights(plist, prec)
    c2f = ops.c2f_weights(sd["progress"].to(d), opt.barf_c2f, d)
    use_noise = bool(opt.nerf.density_noise_reg) and mode == "train"
    out = ops.nerf_pass(center.to(d), dirs.to(d), t.to(d), noise[0].to(d) if use_noise else None,
                        float(opt.nerf.density_noise_reg or 0.0), bool(opt.nerf.setbg_opaque or opt.mask_img), prec, packed, c2f, plist)
    return out


def oracle_forward(opt, sd, center, dirs, t, noise, mode):
    c, r, tt = center[None], dirs[None], t[None, :, :, None]
    rgb_s, dens = O.mlp(opt, sd, O.points_from_depth(c, r, tt), r, mode, noise)
    out = dict(rgb_samples=rgb_s, density_samples=dens)
    out.update(O.composite(opt, r, rgb_s, dens, tt))
    return out


CASES = [
    ("plain", dict(), "val"),
    ("noise_bg", dict(nerf=dict(density_noise_reg=True, setbg_opaque=True)), "train"),
    ("c2f", dict(barf_c2f=[0.4, 0.7]), "train"),
]


@pytest.mark.parametrize("prec", [L.PREC_FP32, L.PREC_BF16, L.PREC_X3], ids=["fp32", "bf16", "bf16x3"])
@pytest.mark.parametrize("tag,over,mode", CASES, ids=[c[0] for c in CASES])
def test_pass_forward(prec, tag, over, mode):
    R, N = 70, 24            # 1680 rows: not a multiple of the 256/128-row workgroup tile
    opt = small_opt(**over)
    sd = make_state_dict(opt, 5, progress=0.55 if opt.barf_c2f is not None else None)
    center, dirs, jitter, noise = make_scene(R, N, 1)
    t = O.sample_depth(opt, 1, R, N, [1.2, 5.2], "train", jitter)[0, :, :, 0]
    ref = oracle_forward(opt, sd, center, dirs, t, noise, mode)
    got = run_forward(prec, opt, sd, center, dirs, t, noise, mode)
    tol = TOL[prec]
    errs = {}
    for k, shape in (("rgb_samples", (R, N, 3)), ("density_samples", (R, N)), ("weights", (R, N)), ("rgb", (R, 3)), ("depth", (R,)),
                     ("opacity", (R,)), ("depth_var", (R,)), ("rgb_var", (R,)), ("all_cumulated", (R,))):
        errs[k] = rel_err(got[k].reshape(shape), ref[k].reshape(shape))
    # rgb_var = sum_i w_i sum_ch (c_i - rgb) is ~0 by construction (weights sum to 1, SURVEY
    # quirk 6): a cancellation residue, so it gets an absolute bound
    errs["rgb_var"] = float((got["rgb_var"].reshape(R).cpu() - ref["rgb_var"].reshape(R)).abs().max())
    print(tag, prec, errs)
    bad = {k: e for k, e in errs.items() if not e < tol}
    assert not bad, bad


def test_sample_fine_matches_oracle():
    R, Nc, Nf = 53, 16, 40
    rs = np.random.RandomState(2)
    w = torch.from_numpy(rs.gamma(0.5, 1.0, size=(R, Nc)).astype(np.float32))
    w = w / w.sum(-1, keepdim=True) * 0.9
    w[3, 9:] = 0
    tc = O.sample_depth(small_opt(), 1, R, Nc, [1.2, 5.2], "train", torch.from_numpy(rs.uniform(0, 1, size=(1, R, Nc, 1)).astype(np.float32)))[0, :, :, 0]
    for grid in (O.det_grid(Nf), torch.from_numpy(rs.uniform(0, 1, size=Nf + 1).astype(np.float32))):
        ref_f = O.sample_pdf(w[None], Nc, Nf, [1.2, 5.2], grid)[0, :, :, 0]
        ref = torch.cat([tc, ref_f], 1).sort(1).values
        u_mid = 0.5 * (grid[:-1] + grid[1:])
        got, got_f = ops.sample_fine(w.to(dev()), tc.to(dev()), u_mid.to(dev()), 1.2, 5.2, want_unsorted=True)
        # a few ulps at t ~ 1..5: the pdf division / cdf rounding differ in the last bit
        np.testing.assert_allclose(got_f.cpu().numpy(), ref_f.numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=0, atol=1e-5)
        assert (got[:, 1:] >= got[:, :-1]).all()


@pytest.mark.parametrize("prec", [L.PREC_FP32, L.PREC_BF16, L.PREC_X3], ids=["fp32", "bf16", "bf16x3"])
@pytest.mark.parametrize("pose", [False, True], ids=["fixed_pose", "pose_grad"])
def test_pass_backward(prec, pose):
    R, N = 41, 16
    opt = small_opt(barf_c2f=[0.4, 0.7], nerf=dict(density_noise_reg=True, setbg_opaque=True))
    sd = make_state_dict(opt, 9, progress=0.62)
    center, dirs, jitter, noise = make_scene(R, N, 4)
    t = O.sample_depth(opt, 1, R, N, [1.2, 5.2], "train", jitter)[0, :, :, 0]
    rs = np.random.RandomState(8)
    lw = {k: torch.from_numpy(rs.uniform(-1, 1, size=s).astype(np.float32))
          for k, s in (("rgb", (R, 3)), ("depth", (R,)), ("opacity", (R,)), ("weights", (R, N)))}
    # oracle
    sdo = {k: v.clone().requires_grad_(k != "progress") for k, v in sd.items()}
    co, do = center.clone().requires_grad_(True), dirs.clone().requires_grad_(True)
    ref = oracle_forward(opt, sdo, co, do, t, noise, "train")
    loss = sum((ref[k].reshape(v.shape) * v).sum() for k, v in lw.items())
    loss.backward()
    # HIP
    d = dev()
    plist = [p.clone().requires_grad_(True) for p in params_list(sd, d)]
    packed = ops.pack_weights(plist, prec)
    c2f = ops.c2f_weights(sd["progress"].to(d), opt.barf_c2f, d)
    cg, dg = center.to(d).requires_grad_(pose), dirs.to(d).requires_grad_(pose)
    got = ops.nerf_pass(cg, dg, t.to(d), noise[0].to(d), 1.0, True, prec, packed, c2f, plist)
    loss_g = sum((got[k] * v.to(d)).sum() for k, v in lw.items())
    loss_g.backward()
    tol = GTOL[prec]
    # bf16x3 forward errors (~2e-5) flip the occasional ReLU whose pre-activation is ~0: one such flip changes a
    # whole row's contribution to the upstream gradients (~1/rows in max norm), so it is measured in relative L2
    metric = rel_err if prec == L.PREC_FP32 else rel_l2
    errs = {"loss": abs(loss_g.item() - loss.item()) / abs(loss.item())}
    i = 0
    for n in L.PARAM_NAMES:
        for k in ("weight", "bias"):
            errs[f"{n}.{k}"] = metric(plist[i].grad, sdo[f"{n}.{k}"].grad)
            i += 1
    if pose:
        # bf16: the encoding gradient multiplies bf16 noise by 2^k pi -- reported, loosely bounded
        errs["d_center"] = metric(cg.grad, co.grad) * (0.25 if prec == L.PREC_BF16 else 1.0)
        errs["d_dir"] = metric(dg.grad, do.grad) * (0.25 if prec == L.PREC_BF16 else 1.0)
    print(prec, pose, errs)
    bad = {k: e for k, e in errs.items() if not e < tol}
    assert not bad, bad


@pytest.mark.parametrize("mode", ["pixels_shared", "pixels_per_image", "idx_shared", "idx_per_image", "all"])
def test_ray_gen_matches_camera_restatement(mode):
    """Fused ray generation against the oracle's restatement of
    camera.get_center_and_ray[_at_pixels] (camera.py:347-416): values and d pose."""
    from sparf_amd import ops
    from tests.golden.recipe import ring_cameras
    B, H, W, N = 3, 30, 40, 257
    pose, intr = ring_cameras(B, H=H, W=W)
    rs = np.random.RandomState(5)
    px = idx = None
    if mode == "pixels_shared":
        px = torch.from_numpy(rs.uniform(0, [W, H], size=(N, 2)).astype(np.float32))
    elif mode == "pixels_per_image":
        px = torch.from_numpy(rs.uniform(0, [W, H], size=(B, N, 2)).astype(np.float32))
    elif mode == "idx_shared":
        idx = torch.from_numpy(rs.randint(0, H * W, size=(N,)))
    elif mode == "idx_per_image":
        idx = torch.from_numpy(rs.randint(0, H * W, size=(B, N)))
    else:
        idx = torch.arange(H * W)
    gc = torch.from_numpy(rs.normal(size=(B, (H * W if mode == "all" else N), 3)).astype(np.float32))
    gr = torch.from_numpy(rs.normal(size=gc.shape).astype(np.float32))

    p_ref = pose.clone().requires_grad_(True)
    px_ref = px.clone().requires_grad_(True) if px is not None else None
    if px is not None:
        c_ref, r_ref = O.rays_at_pixels(p_ref, intr, px_ref if px.dim() == 3 else px_ref[None].expand(B, -1, -1))
    else:
        c_ref, r_ref = O.rays_at_index(p_ref, intr, H, W, idx)
    ((c_ref * gc).sum() + (r_ref * gr).sum()).backward()

    p_hip = pose.clone().to(dev()).requires_grad_(True)
    # pixel coordinates may carry a gradient too: the depth-consistency loss renders at projections of points back-projected with
    # a rendered depth (depth_cons_loss.py:199-201, 254-262 -> :291) and the reference's ray generation is plain autograd
    px_hip = px.to(dev()).requires_grad_(True) if px is not None else None
    c, r = ops.ray_gen(p_hip, intr.to(dev()), pixels=px_hip, ray_idx=idx.to(dev()) if idx is not None else None, width=W)
    ((c * gc.to(dev())).sum() + (r * gr.to(dev())).sum()).backward()
    if px is not None:
        assert px_hip.grad is not None and px_hip.grad.shape == px_ref.grad.shape
        assert float((px_hip.grad.cpu() - px_ref.grad).abs().max()) <= 1e-5 * float(px_ref.grad.abs().max())
    assert c.shape == c_ref.shape and r.shape == r_ref.shape
    assert float((c.detach().cpu() - c_ref.detach()).abs().max()) <= 2e-6 * float(c_ref.detach().abs().max())
    assert float((r.detach().cpu() - r_ref.detach()).abs().max()) <= 2e-6 * float(r_ref.detach().abs().max())
    gref = p_ref.grad
    assert float((p_hip.grad.cpu() - gref).abs().max()) <= 1e-4 * float(gref.abs().max())


@pytest.mark.parametrize("huber", [False, True])
@pytest.mark.parametrize("fine", [False, True])
def test_photometric_loss_matches_reference_formulas(huber, fine):
    """ops.photometric_loss against BaseLoss.MSE_loss / huber_loss as written in
    base_losses.py:151-156 and summed in :303-311 (values and gradients)."""
    from sparf_amd import ops
    rs = np.random.RandomState(7)
    n = 1237
    tgt = torch.from_numpy(rs.uniform(size=(2, n, 3)).astype(np.float32)).to(dev())
    a = torch.from_numpy(rs.uniform(-0.5, 1.5, size=(2, n, 3)).astype(np.float32)).to(dev())
    b = torch.from_numpy(rs.uniform(-0.5, 1.5, size=(2, n, 3)).astype(np.float32)).to(dev())

    def ref_loss(p):
        if huber:
            return torch.nn.functional.huber_loss(p, tgt, reduction="mean", delta=0.5) * 2.
        e = (p.contiguous() - tgt) ** 2
        return e.sum() / (e.nelement() + 1e-6)

    a1, b1 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = ref_loss(a1) + (ref_loss(b1) if fine else 0.0)
    (ref * 3.0).backward()
    a2, b2 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ours = ops.photometric_loss(a2, tgt, rgb_fine=b2 if fine else None, huber=huber)
    (ours * 3.0).backward()
    assert abs(float(ours) - float(ref)) <= 2e-6 * abs(float(ref))
    assert torch.allclose(a2.grad, a1.grad, rtol=1e-5, atol=1e-9)
    if fine:
        assert torch.allclose(b2.grad, b1.grad, rtol=1e-5, atol=1e-9)


def test_pass_backward_far_samples():
    """Inverse-depth sampling (renderer.py:413-416) puts samples at t up to 1e8, intervals up to 1e8 and
    the closing interval at 1e10: d loss / d sigma_j = delta_j |ray| (T_{j+1} q_j - sum_{k>j} w_k q_k)
    multiplies whatever residue the suffix sum carries by those factors.  Regression test of the
    far-end-first suffix accumulation in composite_bwd_kernel (a total - prefix form, even in fp64, left
    1e-16 |total| ~ 1e-8 behind the last sample under a depth loss and turned it into O(100) gradients:
    3e-2 relative error on the first-layer weights at BASELINE config 3)."""
    from tests import scale_cases as S
    R, N = 96, 64
    opt = small_opt(nerf=dict(depth=dict(param="inverse", range=[1, 0]), sample_intvs=N))
    sd = make_state_dict(opt, 103)
    center, dirs, jitter, _ = make_scene(R, N, 11)
    t = O.sample_depth(opt, 1, R, N, [1, 0], "train", jitter)            # [1,R,N,1], up to ~1e8
    assert float(t.max()) > 1e3
    rs = np.random.RandomState(12)
    lw = {"depth": torch.from_numpy(rs.uniform(-1, 1, size=(1, R, 1)).astype(np.float32)),
          "rgb": torch.from_numpy(rs.uniform(-1, 1, size=(1, R, 3)).astype(np.float32))}
    _, gref, _, _ = S.referee(opt, sd, sd, center[None], dirs[None], t, None, None, None, lw, "train", chunk=R, want_ray_grad=False)
    _, g32, _, _ = S.referee(opt, sd, sd, center[None], dirs[None], t, None, None, None, lw, "train", chunk=R, want_ray_grad=False, dtype=torch.float32)
    d = dev()
    plist = [p.clone().requires_grad_(True) for p in params_list(sd, d)]
    packed = ops.pack_weights(plist, L.PREC_FP32)
    c2f = ops.c2f_weights(sd["progress"].to(d), None, d)
    got = ops.nerf_pass(center.to(d), dirs.to(d), t[0, :, :, 0].to(d), None, 0.0, False, L.PREC_FP32, packed, c2f, plist)
    loss = (got["depth"] * lw["depth"].reshape(R).to(d)).sum() + (got["rgb"] * lw["rgb"].reshape(R, 3).to(d)).sum()
    loss.backward()
    names = [f"{n}.{k}" for n in L.PARAM_NAMES for k in ("weight", "bias")]
    ours = max(rel_l2(p.grad, gref["nerf"][k]) for k, p in zip(names, plist))
    ref32 = max(rel_l2(g32["nerf"][k], gref["nerf"][k]) for k in names)
    print("far samples: worst parameter-gradient relative L2 vs float64 referee: HIP fp32", ours, "reference fp32", ref32)
    assert ours < max(5e-3, 5 * ref32)


def test_large_pass_beyond_the_old_2GiB_limit():
    """One training pass of 2.36 M sample rows (12 288 rays x 192, bf16: a 10.7 GB save area and a 10.6 GB gradient area).
    Round 2 addressed the saved buffers with 32-bit byte offsets (rows * 320 * 4 < 2^31, ~1.6 M rows) and cut such a batch
    into chunks; the tile-block-major areas (layout.h) have no such limit.  Checked against the same rays run as three
    4096-ray passes: outputs bit-identical (rays are independent), ray gradients bit-identical, parameter gradients equal
    to summation order."""
    R, N, prec = 12288, 192, L.PREC_BF16
    lib = L.load()
    assert lib.sparf_save_bytes(prec, R * N) > (1 << 33)                     # > 8 GiB: far beyond a 32-bit offset
    opt = small_opt(nerf=dict(setbg_opaque=False))
    sd = make_state_dict(opt, 21)
    d = dev()
    g = torch.Generator().manual_seed(5)
    center = (torch.rand(R, 3, generator=g) - 0.5 + torch.tensor([0.0, 0.0, -3.0])).to(d)
    dirs = (torch.rand(R, 3, generator=g) * 0.6 - 0.3 + torch.tensor([0.0, 0.0, 1.0])).to(d)
    t = (torch.sort(torch.rand(R, N, generator=g), dim=1).values * 4.0 + 1.2).to(d)
    w_rgb, w_depth = torch.rand(R, 3, generator=g).to(d), torch.rand(R, generator=g).to(d)
    c2f = ops.c2f_weights(sd["progress"].to(d), None, d)

    def run(chunks):
        plist = [p.clone().requires_grad_(True) for p in params_list(sd, d)]
        packed = ops.pack_weights(plist, prec)
        cg, dg = center.clone().requires_grad_(True), dirs.clone().requires_grad_(True)
        outs, loss = [], 0.0
        for lo in range(0, R, R // chunks):
            hi = lo + R // chunks
            o = ops.nerf_pass(cg[lo:hi], dg[lo:hi], t[lo:hi], None, 0.0, False, prec, packed, c2f, plist)
            outs.append(o)
            loss = loss + (o["rgb"] * w_rgb[lo:hi]).sum() + (o["depth"] * w_depth[lo:hi]).sum()
        loss.backward()
        cat = {k: torch.cat([o[k] for o in outs]) for k in ("rgb", "depth", "opacity", "weights")}
        return cat, torch.cat([p.grad.reshape(-1) for p in plist]), cg.grad, dg.grad

    one, gp1, dc1, dd1 = run(1)
    three, gp3, dc3, dd3 = run(3)
    for k in one:
        assert torch.equal(one[k], three[k]), k
    assert torch.equal(dc1, dc3) and torch.equal(dd1, dd3)
    assert rel_l2(gp1, gp3) < 1e-5
    torch.cuda.empty_cache()


@pytest.mark.parametrize("pose", [False, True])
@pytest.mark.parametrize("R,N", [(70, 24), (1024, 32), (515, 64), (1536, 64)])
def test_both_geometries_of_the_bf16x3_data_gradient_kernel_are_bit_identical(R, N, pose):
    """The bf16x3 dgrad ships as an 8-wave / 256-row and a 4-wave / 128-row kernel and sparf_pass_backward picks one per launch from the
    row count (api.hip x3_dgrad_waves, round 6).  Pinned through sparf_launch_kernel (3 / 4), both must leave the SAME bytes in the
    workspace -- gradient area (every dY the weight-gradient kernel reads), d point and d view encoding -- on ragged row counts too
    (1 680 rows: neither a multiple of 128 nor of 256; 32 960: a partial last round of either tile size), and so must the launch plan
    the pass itself uses (which = 1): at 98 304 rows on a 256-CU chip that is one full round of 256-row tiles in 8 waves followed by the
    remaining 32 768 rows in 4 waves (api.hip x3_dgrad_rows8)."""
    import ctypes
    lib = L.load()
    d = dev()
    prec = L.PREC_X3
    opt = small_opt(barf_c2f=[0.4, 0.7])
    sd = make_state_dict(opt, 21, progress=0.55)
    center, dirs, jitter, _ = make_scene(R, N, 6)
    t = O.sample_depth(opt, 1, R, N, [1.2, 5.2], "train", jitter)[0, :, :, 0].to(d).contiguous()
    plist = params_list(sd, d)
    packed = ops.pack_weights(plist, prec)
    c2f = ops.c2f_weights(sd["progress"].to(d), opt.barf_c2f, d)
    c, dr = center.to(d).contiguous(), dirs.to(d).contiguous()
    fa, out, save, keep1 = ops.build_pass_fwd(prec, c, dr, t, None, 0.0, False, packed, c2f, True)
    s = L.stream_ptr(d)
    L.check(lib.sparf_pass_forward(ctypes.byref(fa), s), "fwd")
    g = torch.Generator().manual_seed(3)
    grads = (torch.rand(R, 3, generator=g).to(d), torch.rand(R, generator=g).to(d), None, torch.rand(R, N, generator=g).to(d))
    ba, gp, dc, dd, keep2 = ops.build_pass_bwd(prec, c, dr, t, None, 0.0, False, packed, c2f, save, out, grads, pose)
    L.check(lib.sparf_pass_backward(ctypes.byref(ba), s), "bwd")            # fills d sigma / d z of the workspace
    ws = keep2[0]
    images = []
    for which in (3, 4, 1):
        L.check(lib.sparf_launch_kernel(which, ctypes.byref(fa), ctypes.byref(ba), s), "dgrad")
        torch.cuda.synchronize()
        images.append(ws.clone())
    assert torch.equal(images[0], images[1]), int((images[0] != images[1]).sum())
    assert torch.equal(images[0], images[2]), int((images[0] != images[2]).sum())
    assert int((images[0] != 0).sum()) > ws.numel() // 8                     # (the comparison is of real content)
