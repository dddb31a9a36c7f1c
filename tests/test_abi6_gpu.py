"""C ABI 6 (round 5), against the oracle's autograd:
  * EVERY output of a composite carries its gradient -- depth_var, rgb_var, all_cumulated, density_samples, rgb_samples next to
    rgb / depth / opacity / weights: NeRF.composite is plain autograd in the reference (frequency_nerf.py:317-338); rounds 1-4
    marked them non-differentiable and a loss on depth_var silently got zero (VERDICT r04 missing-3);
  * NeRF.composite as a free function of caller-built per-sample values (VERDICT r04 missing-4), outputs on the reference's golden
    vectors and gradients against the oracle;
  * Graph.render as ONE autograd node (ops.RenderFn) = the pass-by-pass path bit for bit, outputs and gradients;
  * far tiles by value are refused for main precisions whose workgroup tile differs from the far kernel's.
Run with `pytest -m gpu`."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from sparf_amd import lib as L
from sparf_amd import ops
from sparf_amd.renderer import Graph
from tests.golden.recipe import make_state_dict, small_opt

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def dev():
    return torch.device("cuda:0")


def rel(a, b, floor=0.0):
    """max|a - b| / max(max|b|, floor).  floor: for quantities that are rounding noise around 0 -- rgb_var = (sum_ch rgb)(1 - opacity)
    with opacity == 1 up to 1e-7 (the last sample absorbs the residual transmittance) -- whose RELATIVE distance means
    nothing; their scale is the colour's, O(1)"""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), floor, 1e-30))


def out_rel(k, a, b):
    return rel(a, b, floor=1.0 if k.startswith("rgb_var") else 0.0)


def build(opt, seed, progress=None):
    g = Graph(opt, dev())
    g.nerf.load_state_dict(make_state_dict(opt, seed, progress))
    if opt.nerf.fine_sampling:
        g.nerf_fine.load_state_dict(make_state_dict(opt, seed + 1, progress))
    return g


def scene(B, R, N, seed):
    rs = np.random.RandomState(seed)
    c = T(rs.uniform(-0.3, 0.3, size=(B, R, 3)).astype(np.float32)) + torch.tensor([0.0, 0.0, -3.0])
    r = T(rs.uniform(-0.3, 0.3, size=(B, R, 3)).astype(np.float32)) + torch.tensor([0.0, 0.0, 1.0])
    t = T(np.sort(rs.uniform(1.2, 5.2, size=(B, R, N, 1)), axis=2).astype(np.float32))
    return c, r, t, rs


ALL_KEYS = ("rgb", "depth", "opacity", "weights", "depth_var", "rgb_var", "all_cumulated", "density_samples", "rgb_samples")


@pytest.mark.parametrize("white_bg", [False, True])
@pytest.mark.parametrize("key", ["depth_var", "rgb_var", "all_cumulated", "density_samples", "rgb_samples", "all"])
def test_every_pass_output_is_differentiable(key, white_bg):
    """one fused pass (fp32 mode: the comparison is of the formulas), a loss on ONE of the outputs the earlier rounds
    left without gradient (and on all nine together): parameter and ray gradients against the oracle's autograd"""
    opt = small_opt(nerf=dict(setbg_opaque=white_bg), hip=dict(precision="fp32"))
    graph = build(opt, 21)
    B, R, N = 2, 13, 24
    c, r, t, rs = scene(B, R, N, 4)
    sd = {k: v.detach().cpu().clone().requires_grad_(k != "progress") for k, v in graph.nerf.state_dict().items()}
    co, ro = c.clone().requires_grad_(True), r.clone().requires_grad_(True)
    ref = O.pass_fixed(opt, sd, co, ro, t, mode="val")
    cg, rg = c.to(dev()).requires_grad_(True), r.to(dev()).requires_grad_(True)
    out = graph.nerf.render_pass(opt, cg, rg, t.to(dev()), mode="val")
    keys = ALL_KEYS if key == "all" else (key,)
    coef = {k: T(rs.normal(size=tuple(ref[k].shape)).astype(np.float32)) for k in keys}
    for k in ALL_KEYS:
        assert out[k].requires_grad, f"{k} must carry a gradient (frequency_nerf.py:317-338 is plain autograd)"
        assert out_rel(k, out[k], ref[k]) < 1e-4, (k, out_rel(k, out[k], ref[k]))
    sum((ref[k] * coef[k]).sum() for k in keys).backward()
    sum((out[k] * coef[k].to(dev())).sum() for k in keys).backward()
    for name, p in graph.nerf.named_parameters():
        if name == "progress":
            continue
        ref_g = sd[name].grad
        if ref_g is None or float(ref_g.abs().max()) == 0.0:       # (depth_var, all_cumulated, density do not depend on the colour branch)
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, (key, name)
            continue
        assert p.grad is not None, (key, name)
        # rgb_var = (sum_ch rgb)(1 - opacity) with opacity == 1: its gradient is d opacity times a constant, i.e. cancellation noise of
        # order 1e-7 in the oracle and here -- an absolute comparison far below the other keys' gradients (~1e-2) is what is left to check
        if key == "rgb_var":
            assert float((p.grad.detach().cpu() - ref_g).abs().max()) < 1e-6, (key, name)
            continue
        assert rel(p.grad, ref_g) < 2e-3, (key, name, rel(p.grad, ref_g))
    if key == "rgb_var":      # (rounding noise on both sides, 1e-5 in the ray gradients: sum_i dw_i = 0 analytically, 1e-7 T in fp32, times the MLP's Jacobian)
        assert float((cg.grad.cpu() - co.grad).abs().max()) < 5e-4 and float((rg.grad.cpu() - ro.grad).abs().max()) < 5e-4
        return
    assert rel(cg.grad, co.grad) < 2e-3 and rel(rg.grad, ro.grad) < 2e-3, (rel(cg.grad, co.grad), rel(rg.grad, ro.grad))


def test_render_batch_segments_route_the_new_gradients():
    """the segment table of a batched pass carries the five new upstream gradients per request"""
    opt = small_opt(hip=dict(precision="fp32"))
    graph = build(opt, 23)
    graph.train()
    H, W = 20, 30
    rs = np.random.RandomState(3)
    pose = torch.tensor([[[1.0, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 3.0]]], device=dev())
    intr = torch.tensor([[[25.0, 0, W / 2], [0, 25.0, H / 2], [0, 0, 1]]], device=dev())
    reqs = [dict(pose=pose, H=H, W=W, intr=intr, pixels=T(rs.uniform(0, [W, H], size=(1, n, 2)).astype(np.float32)).to(dev()), depth_range=[1.5, 4.5], mode="val")
            for n in (17, 40)]
    coefs = [torch.randn(1, n, 1, device=dev()) for n in (17, 40)]

    def loss_of(rets):
        return sum((r.depth_var * c).sum() + (r.all_cumulated_fine[..., None] * c).sum() + 0.3 * (r.rgb_var_fine * c).sum() for r, c in zip(rets, coefs))

    graph.zero_grad(set_to_none=True)
    loss_of(graph.render_batch(opt, reqs)).backward()
    g_batch = {n: p.grad.clone() for n, p in graph.named_parameters() if p.grad is not None}
    graph.zero_grad(set_to_none=True)
    loss_of([graph.render(opt, q["pose"], H=H, W=W, intr=intr, pixels=q["pixels"], depth_range=q["depth_range"], mode="val") for q in reqs]).backward()
    assert all(float(g_batch[f"{net}.mlp_feat.0.weight"].abs().max()) > 0 for net in ("nerf", "nerf_fine"))       # (the colour branch of the coarse
    #  network legitimately gets none: depth_var does not depend on the colours)
    for n, p in graph.named_parameters():
        if p.grad is not None:
            assert rel(g_batch[n], p.grad) < 1e-5, (n, rel(g_batch[n], p.grad))


def test_standalone_composite_on_reference_golden_vectors(golden):
    """NeRF.composite on a dictionary the caller built (no forward_samples): the reference's own golden composite fixture
    (tests/golden/composite.npz, made by the reference module) through the public method"""
    g = golden("composite")
    graph = build(small_opt(), 33)
    ray, rgbs, dens, t = (T(g[k]).to(dev()) for k in ("in_ray", "in_rgb_s", "in_density", "in_t"))
    for tag, bg in (("plain", False), ("bg", True)):
        opt = small_opt(nerf=dict(setbg_opaque=bg))
        out = graph.nerf.composite(opt, ray, dict(rgb_samples=rgbs.clone(), density_samples=dens.clone()), t)
        for k in ("rgb", "rgb_var", "depth", "depth_var", "opacity", "weights", "all_cumulated"):
            ref = g[f"out_{tag}_{k}"]
            assert tuple(out[k].shape) == tuple(ref.shape), (tag, k)
            assert out_rel(k, out[k], T(ref)) < 2e-5, (tag, k, out_rel(k, out[k], T(ref)))


@pytest.mark.parametrize("white_bg", [False, True])
def test_standalone_composite_gradients_match_oracle(white_bg):
    opt = small_opt(nerf=dict(setbg_opaque=white_bg))
    graph = build(opt, 35)
    B, R, N = 2, 9, 20
    c, r, t, rs = scene(B, R, N, 8)
    dens = T(rs.uniform(0.0, 3.0, size=(B, R, N)).astype(np.float32))
    rgbs = T(rs.uniform(0.0, 1.0, size=(B, R, N, 3)).astype(np.float32))
    ro, do, co = r.clone().requires_grad_(True), dens.clone().requires_grad_(True), rgbs.clone().requires_grad_(True)
    ref = O.composite(opt, ro, co, do, t)
    rg, dg, cg = r.to(dev()).requires_grad_(True), dens.to(dev()).requires_grad_(True), rgbs.to(dev()).requires_grad_(True)
    out = graph.nerf.composite(opt, rg, dict(rgb_samples=cg, density_samples=dg), t.to(dev()))
    keys = ("rgb", "depth", "opacity", "weights", "depth_var", "rgb_var", "all_cumulated")
    coef = {k: T(rs.normal(size=tuple(ref[k].shape)).astype(np.float32)) for k in keys}
    for k in keys:
        assert tuple(out[k].shape) == tuple(ref[k].shape) and out_rel(k, out[k], ref[k]) < 2e-5, (k, out_rel(k, out[k], ref[k]))
    sum((ref[k] * coef[k]).sum() for k in keys).backward()
    sum((out[k] * coef[k].to(dev())).sum() for k in keys).backward()
    assert rel(dg.grad, do.grad) < 1e-4, rel(dg.grad, do.grad)
    assert rel(cg.grad, co.grad) < 1e-4, rel(cg.grad, co.grad)
    assert rel(rg.grad, ro.grad) < 1e-4, rel(rg.grad, ro.grad)
    with pytest.raises(L.SparfError):
        graph.nerf.composite(opt, rg, dict(rgb_samples=cg, density_samples=dg), t.to(dev()).requires_grad_(True))


@pytest.mark.parametrize("case", ["metric_noise_c2f", "inverse_far_rows", "gated", "val"])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_fused_render_equals_pass_by_pass(case, precision, monkeypatch):
    """Graph.render as one autograd node (ops.RenderFn) against the pass-by-pass path (opt.hip.fused_render = False): the same
    kernels in the same order on the same draws -- outputs, parameter gradients, pose gradient BIT-identical"""
    over = dict(metric_noise_c2f=dict(barf_c2f=[0.2, 0.6], nerf=dict(density_noise_reg=True, setbg_opaque=True)),
                inverse_far_rows=dict(nerf=dict(depth=dict(param="inverse", range=[1, 0]))),
                gated=dict(nerf=dict(ratio_start_fine_sampling_at_x=0.5), max_iter=1000), val=dict())[case]
    mode = "val" if case == "val" else "train"
    it = 10 if case == "gated" else 900
    H, W, B, R = 24, 32, 2, 37
    rs = np.random.RandomState(6)
    pose = torch.tensor([[[1.0, 0, 0, 0.1], [0, 1, 0, 0], [0, 0, 1, 3.0]], [[1.0, 0, 0, -0.2], [0, 1, 0, 0.1], [0, 0, 1, 3.2]]])
    intr = torch.tensor([[[28.0, 0, W / 2], [0, 28.0, H / 2], [0, 0, 1]]] * B)
    pixels = T(rs.uniform(0, [W, H], size=(B, R, 2)).astype(np.float32))
    res = {}
    for fused in (True, False):
        opt = small_opt(hip=dict(precision=precision, fused_render=fused), **over)
        graph = build(opt, 41, progress=0.45 if "c2f" in case else None)
        graph.train()
        torch.manual_seed(5)
        torch.cuda.manual_seed(5)
        pg = pose.to(dev()).requires_grad_(True)
        calls = []
        if fused:
            real = ops.RenderFn.apply
            monkeypatch.setattr(ops.RenderFn, "apply", lambda *a: (calls.append(1), real(*a))[1])
        ret = graph.render(opt, pg, H=H, W=W, intr=intr.to(dev()), pixels=pixels.to(dev()), depth_range=[1, 0] if "inverse" in case else [1.5, 4.5],
                           iter=it, mode=mode)
        if fused:
            monkeypatch.undo()
            assert calls == [1], "the fused path did not run"
        keys = [k for k in ret.keys() if torch.is_tensor(ret[k]) and ret[k].dtype.is_floating_point]
        loss = sum((ret[k] * torch.linspace(0.5, 1.5, ret[k].numel(), device=dev()).view(ret[k].shape)).sum()
                   for k in keys if ret[k].requires_grad and k not in ("origins", "viewdirs"))
        loss.backward()
        res[fused] = (dict((k, ret[k].detach().clone()) for k in keys), {n: p.grad.clone() for n, p in graph.named_parameters() if p.grad is not None},
                      pg.grad.clone())
    assert set(res[True][0]) == set(res[False][0])
    assert ("rgb_fine" in res[True][0]) == (case != "gated")
    for k in res[True][0]:
        assert res[True][0][k].shape == res[False][0][k].shape, k
        assert torch.equal(res[True][0][k], res[False][0][k]), (case, precision, k, rel(res[True][0][k], res[False][0][k]))
    assert set(res[True][1]) == set(res[False][1])
    for n in res[True][1]:
        assert torch.equal(res[True][1][n], res[False][1][n]), (case, precision, n, rel(res[True][1][n], res[False][1][n]))
    assert torch.equal(res[True][2], res[False][2]), rel(res[True][2], res[False][2])


def test_fused_render_skips_the_pass_without_gradient():
    """a loss that reads the fine pass only (corres_loss.py:183 reads depth_fine): the coarse network receives no gradient at all,
    as under the pass-by-pass path, where autograd never calls the coarse pass's backward"""
    opt = small_opt(hip=dict(precision="fp32"))
    graph = build(opt, 43)
    graph.train()
    H, W = 20, 30
    pose = torch.tensor([[[1.0, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 3.0]]], device=dev())
    intr = torch.tensor([[[25.0, 0, W / 2], [0, 25.0, H / 2], [0, 0, 1]]], device=dev())
    ret = graph.render(opt, pose, H=H, W=W, intr=intr, ray_idx=torch.arange(50, device=dev()), depth_range=[1.5, 4.5], iter=5, mode="train")
    ret.depth_fine.sum().backward()
    assert all(p.grad is None for n, p in graph.nerf.named_parameters())
    assert all(p.grad is not None for n, p in graph.nerf_fine.named_parameters() if n != "progress")
    assert all(float(p.grad.abs().max()) > 0 for n, p in graph.nerf_fine.named_parameters() if n.startswith("mlp_feat"))      # (depth does not depend on the colours)


def test_far_tiles_by_value_need_matching_workgroup_tiles():
    """far_count = -1 decides per WORKGROUP tile; the bf16 kernels run 8 waves (256-row tiles), the fp32 far kernel 4
    (128-row tiles): a 256-row tile that straddles the threshold would be half-evaluated.  The C ABI refuses the combination."""
    lib = L.load()
    R, N = 8, 32
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev())
    opt = small_opt()
    graph = build(opt, 3)
    c, d, t = f(R, 3).normal_(), f(R, 3).normal_(), torch.sort(f(R, N).uniform_(1, 5), dim=1).values
    for prec, ok in ((L.PREC_X3, True), (L.PREC_BF16, False)):
        packed, fpacked = graph.nerf.packed(prec), graph.nerf.packed(L.PREC_FP32)
        a, out, _, keep = ops.build_pass_fwd(prec, c, d, t, None, 0.0, False, packed, graph.nerf.band_weights(), False, far=(8.0, L.PREC_FP32, fpacked))
        rc = lib.sparf_pass_forward(ctypes.byref(a), L.stream_ptr(dev()))
        torch.cuda.synchronize()
        assert (rc == 0) == ok, (prec, rc)
