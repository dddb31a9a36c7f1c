"""Lazy batching of back-to-back render calls (round 6; sparf_amd/renderer.py PendingRender): what the UNMODIFIED
correspondence loss does -- `render_image_at_specific_pose_and_rays` twice, reading neither result before both exist
(corres_loss.py:158-166) -- runs as ONE `render_batch`, with the results, the random streams and the gradients of the separate calls."""
import copy

import pytest
import torch

from sparf_amd.config import default_opt
from sparf_amd.edict import EasyDict as edict

pytestmark = pytest.mark.gpu


def _setup(lazy, precision="fp32", noise=False, seed=0, inverse=False):
    from sparf_amd.renderer import Graph
    dev = torch.device("cuda:0")
    depth = dict(param="inverse", range=[1, 0]) if inverse else dict(param="metric")      # inverse depth: the last samples of every ray take the fp32 kernels ("far rows") in bf16x3
    opt = default_opt(nerf=dict(sample_intvs=32, sample_intvs_fine=32, fine_sampling=True, rand_rays=256, density_noise_reg=noise, depth=depth),
                      barf_c2f=[0.1, 0.5], hip=dict(precision=precision, lazy_batch=lazy))
    torch.manual_seed(seed)
    g = Graph(opt, dev)
    g.train()
    g.nerf.progress.data.fill_(0.4)
    g.nerf_fine.progress.data.fill_(0.4)
    H, W = 40, 60
    gen = torch.Generator().manual_seed(5)
    poses = torch.tensor([[[1.0, 0, 0, 0.1], [0, 1, 0, 0], [0, 0, 1, 3.0]], [[1.0, 0, 0, -0.2], [0, 1, 0, 0.05], [0, 0, 1, 3.1]]], device=dev).requires_grad_(True)
    intr = torch.tensor([[50.0, 0, W / 2], [0, 50.0, H / 2], [0, 0, 1]], device=dev)
    px = [(torch.rand(n, 2, generator=gen) * torch.tensor([W - 1.0, H - 1.0])).to(dev) for n in (96, 160)]
    data = edict(depth_range=torch.tensor([[1.5, 4.5]] * 2, device=dev))
    return g, opt, data, poses, intr, px, H, W


def _pair(g, opt, data, poses, intr, px, H, W, seed=3):
    torch.manual_seed(seed)
    a = g.render_image_at_specific_pose_and_rays(opt, data, poses[0], intr, H, W, iter=10, pixels=px[0], mode="train")
    b = g.render_image_at_specific_pose_and_rays(opt, data, poses[1], intr, H, W, iter=10, pixels=px[1], mode="train")
    return a, b


def _loss(a, b):
    return (a.rgb_fine.mean() + 0.3 * a.depth.mean() + b.rgb.sum() * 1e-2 + 0.5 * b.depth_fine.mean() + 0.1 * b.opacity_fine.mean())


@pytest.mark.parametrize("noise,inverse", [(False, False), (True, False), (False, True)])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_two_back_to_back_calls_run_as_one_batch_with_the_separate_calls_results(precision, noise, inverse):
    from sparf_amd.renderer import PendingRender
    ref = _setup(False, precision, noise, inverse=inverse)
    a0, b0 = _pair(*ref)
    assert not isinstance(a0, PendingRender)
    _loss(a0, b0).backward()
    g0, poses0 = ref[0], ref[3]

    lz = _setup(True, precision, noise, inverse=inverse)
    g1, poses1 = lz[0], lz[3]
    a1, b1 = _pair(*lz)
    assert isinstance(a1, PendingRender) and isinstance(b1, PendingRender)
    assert g1.lazy_stats == dict(batches=0, requests=0) and g1._pending is not None, "nothing launched before the first read"
    assert dict.__contains__(a1, "ray_idx") and not dict.__contains__(a1, "rgb")          # the wrapper's own write (renderer.py:188) launched nothing
    assert "rgb_fine" in a1.keys()                                                          # the loss code's first read (corres_loss.py:170)
    assert g1.lazy_stats == dict(batches=1, requests=2) and g1._pending is None
    for k in ("rgb", "rgb_fine", "depth", "depth_fine", "opacity", "opacity_fine", "weights_fine", "t_fine", "all_cumulated"):
        assert torch.equal(a1[k], a0[k]) and torch.equal(b1[k], b0[k]), k                   # same kernels on the same draws: bit-identical
    _loss(a1, b1).backward()
    assert torch.allclose(poses1.grad, poses0.grad, rtol=2e-4, atol=1e-7)
    for (n0, p0), (n1, p1) in zip(g0.named_parameters(), g1.named_parameters()):
        if n0.endswith("progress"):
            continue
        rel = float((p1.grad - p0.grad).norm() / (p0.grad.norm() + 1e-30))
        assert rel <= (2e-5 if precision == "fp32" else 2e-3), (n0, rel)                     # summation order of the row ranges (bf16x3: bf16 dY rounding of other tiles' sums)


def test_a_single_deferred_call_is_the_eager_call_bit_for_bit():
    ref = _setup(False, "bf16x3", True)
    g0, opt0, data0, poses0, intr0, px0, H, W = ref
    torch.manual_seed(9)
    a0 = g0.render_image_at_specific_pose_and_rays(opt0, data0, poses0[0], intr0, H, W, iter=10, pixels=px0[0], mode="train")
    (a0.rgb_fine.sum() + a0.depth.sum()).backward()
    g1, opt1, data1, poses1, intr1, px1, H, W = _setup(True, "bf16x3", True)
    torch.manual_seed(9)
    a1 = g1.render_image_at_specific_pose_and_rays(opt1, data1, poses1[0], intr1, H, W, iter=10, pixels=px1[0], mode="train")
    after = torch.rand(3)                           # the RNG stream behind the call is the eager call's: every draw was taken at call time
    torch.manual_seed(9)
    g0.render_image_at_specific_pose_and_rays(opt0, data0, poses0[0].detach(), intr0, H, W, iter=10, pixels=px0[0], mode="train")
    assert torch.equal(after, torch.rand(3))
    (a1.rgb_fine.sum() + a1.depth.sum()).backward()
    assert g1.lazy_stats == dict(batches=1, requests=1)
    assert torch.equal(a1.rgb_fine, a0.rgb_fine) and torch.equal(poses1.grad, poses0.grad)
    for (n0, p0), (n1, p1) in zip(g0.named_parameters(), g1.named_parameters()):
        if not n0.endswith("progress"):
            assert torch.equal(p0.grad, p1.grad), n0


def test_what_launches_a_deferred_batch_and_what_is_never_deferred():
    from sparf_amd.renderer import PendingRender
    g, opt, data, poses, intr, px, H, W = _setup(True, "fp32")
    # (i) an optimiser step: the deferred call must see the weights of the iteration that issued it
    torch.manual_seed(4)
    a = g.render_image_at_specific_pose_and_rays(opt, data, poses[0], intr, H, W, iter=10, pixels=px[0], mode="train")
    assert g._pending is not None
    twin = copy.deepcopy(g)                          # (a Graph with an open batch deep-copies: the copy starts without one)
    optim = torch.optim.SGD([p for n, p in g.named_parameters() if not n.endswith("progress")], lr=0.5)
    for p in optim.param_groups[0]["params"]:
        p.grad = torch.ones_like(p)
    optim.step()
    assert g._pending is None and g.lazy_stats["batches"] == 1, "the step hook launched the pending batch first"
    assert dict.__contains__(a, "rgb")
    torch.manual_seed(4)
    a_twin = twin.render_image_at_specific_pose_and_rays(opt, data, poses[0], intr, H, W, iter=10, pixels=px[0], mode="train")
    assert torch.equal(a.rgb_fine, a_twin.rgb_fine), "rendered with the weights from BEFORE the step"
    torch.manual_seed(4)
    a_after = g.render_image_at_specific_pose_and_rays(opt, data, poses[0], intr, H, W, iter=10, pixels=px[0], mode="train")
    assert not torch.equal(a.rgb_fine, a_after.rgb_fine)
    # (ii) another entry point of the renderer
    b = g.render_image_at_specific_pose_and_rays(opt, data, poses[0], intr, H, W, iter=10, pixels=px[0], mode="train")
    assert isinstance(b, PendingRender) and g._pending is not None
    g.render(opt, poses.detach(), H=H, W=W, intr=intr, ray_idx=torch.arange(16, device=poses.device), depth_range=[1.5, 4.5], iter=10, mode="train")
    assert g._pending is None
    # (iii) a different iteration closes the open batch
    c = g.render_image_at_specific_pose_and_rays(opt, data, poses[0], intr, H, W, iter=10, pixels=px[0], mode="train")
    d = g.render_image_at_specific_pose_and_rays(opt, data, poses[0], intr, H, W, iter=11, pixels=px[0], mode="train")
    assert c.__dict__.get("_lz") is None and d.__dict__.get("_lz") is not None
    g.flush_pending()
    assert d.__dict__.get("_lz") is None and dict.__contains__(d, "rgb")
    # never deferred: inference modes, no_grad, the whole-image path
    e = g.render_image_at_specific_pose_and_rays(opt, data, poses[0], intr, H, W, iter=10, pixels=px[0], mode="val")
    with torch.no_grad():
        f = g.render_image_at_specific_pose_and_rays(opt, data, poses[0], intr, H, W, iter=10, pixels=px[0], mode="train")
    assert not isinstance(e, PendingRender) and not isinstance(f, PendingRender)
