"""The renderer as its callers use it (Appendix C items 4-6), on the GPU:

  * the two loss-facing wrappers render_image_at_specific_pose_and_rays /
    render_up_to_maxdepth_at_specific_pose_and_rays (corres_loss.py:158-166,
    depth_cons_loss.py:192,267,291) with (3,4) and (L,3,4) poses, pixel lists and ray indices,
    metric and inverse depth, against the oracle;
  * the public sub-APIs NeRF.forward / forward_samples + composite / positional_encoding /
    FrequencyEmbedder / Graph.sample_depth_from_pdf on the REFERENCE's golden vectors;
  * a caller replay: the render-call sequence of one SPARF training iteration with the grad-mode
    toggles the trainers apply around it (base.py:181-195, depth_cons_loss.py:266, eval.py:31);
  * render_batch against the oracle (not against separate HIP calls);
  * progress.data.fill_ between two renders (nerf_trainer.py:273-275) takes effect immediately.
Run with `pytest -m gpu`."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from sparf_amd.edict import EasyDict as edict
from sparf_amd.renderer import Graph
from tests.golden.recipe import make_state_dict, ring_cameras, small_opt

pytestmark = pytest.mark.gpu
T = torch.from_numpy
COARSE_TOL, FINE_TOL = 5e-4, 3e-2        # end-to-end tolerances of tests/test_graph_gpu.py (conditioning of the resampled depths)


def dev():
    return torch.device("cuda:0")


def build(opt, seed, progress=None):
    g = Graph(opt, dev())
    g.nerf.load_state_dict(make_state_dict(opt, seed, progress))
    if opt.nerf.fine_sampling:
        g.nerf_fine.load_state_dict(make_state_dict(opt, seed + 1, progress))
    return g


def sds(g):
    return ({k: v.detach().cpu() for k, v in g.nerf.state_dict().items()}, {k: v.detach().cpu() for k, v in g.nerf_fine.state_dict().items()})


def max_rel(a, b):
    a, b = a.detach().double().cpu(), torch.as_tensor(b).detach().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def data_dict(B, H, W, pose, intr, rng=(1.2, 5.2)):
    return edict(idx=torch.arange(B), image=torch.zeros(B, 3, H, W, device=dev()), intr=intr.to(dev()), pose=pose.to(dev()),
                 depth_range=torch.tensor([list(rng)] * B, dtype=torch.float32, device=dev()))


def check_render(ret, ref, keys_c=("rgb", "depth", "opacity", "weights", "all_cumulated"), fine=True, scale=1.0):
    for k in keys_c:
        assert tuple(ret[k].shape) == tuple(ref[k].shape), (k, ret[k].shape, ref[k].shape)
        assert max_rel(ret[k], ref[k]) < COARSE_TOL * scale, (k, max_rel(ret[k], ref[k]))
    if fine:
        for k in keys_c:
            assert max_rel(ret[k + "_fine"], ref[k + "_fine"]) < FINE_TOL, (k + "_fine", max_rel(ret[k + "_fine"], ref[k + "_fine"]))


@pytest.mark.parametrize("param", ["metric", "inverse"])
@pytest.mark.parametrize("sel", ["pixels", "ray_idx"])
@pytest.mark.parametrize("nposes", [0, 2], ids=["pose3x4", "poseLx3x4"])
def test_render_image_at_specific_pose_and_rays(param, sel, nposes):
    """renderer.py:142-190: a (3,4) pose (result keeps a leading 1, consumers .squeeze(0)) or an
    (L,3,4) stack; depth range from opt.nerf.depth.range when inverse, else data_dict.depth_range[0]."""
    H, W, B, N = 14, 18, 3, 37
    over = dict(nerf=dict(depth=dict(param=param, range=[1, 0]), rand_rays=64))
    opt = small_opt(**over)
    graph = build(opt, 61)
    pose, intr = ring_cameras(B, H=H, W=W)
    data = data_dict(B, H, W, pose, intr)
    rs = np.random.RandomState(6)
    p_in, k_in = (pose[1], intr[1]) if nposes == 0 else (pose[:nposes], intr[:nposes])
    p_b, k_b = (p_in[None], k_in[None]) if nposes == 0 else (p_in, k_in)
    if sel == "pixels":
        px = T(rs.uniform(0, [W - 1, H - 1], size=(N, 2)).astype(np.float32))
        kw = dict(pixels=px.to(dev()))
        center, ray = O.rays_at_pixels(p_b, k_b, px[None].expand(len(p_b), -1, -1))
    else:
        idx = T(rs.permutation(H * W)[:N])
        kw = dict(ray_idx=idx.to(dev()))
        center, ray = O.rays_at_index(p_b, k_b, H, W, idx)
    with torch.no_grad():
        ret = graph.render_image_at_specific_pose_and_rays(opt, data, p_in.to(dev()), k_in.to(dev()), H, W, iter=7, mode="val", **kw)
        rng = [1, 0] if param == "inverse" else [float(np.float32(1.2)), float(np.float32(5.2))]
        ref = O.render(opt, *sds(graph), center, ray, rng if param == "inverse" else [torch.tensor(1.2), torch.tensor(5.2)], mode="val", it=7)
    assert ret.rgb.shape == (len(p_b), N, 3) and ret.depth.shape == (len(p_b), N, 1)
    assert (ret.ray_idx is None) == (sel == "pixels")
    if sel == "ray_idx":
        assert torch.equal(ret.ray_idx.cpu(), idx)
    assert torch.equal(ret.t.cpu(), ref["t"])
    check_render(ret, ref, scale=4.0 if param == "inverse" else 1.0)


@pytest.mark.parametrize("sel", ["pixels", "ray_idx"])
def test_render_up_to_maxdepth_at_specific_pose_and_rays(sel):
    """renderer.py:460-502 as depth_cons_loss.py:266-273 calls it: under no_grad, (3,4) pose, per-ray
    depth_max [N]; consumers read all_cumulated(_fine).  depth_min = data_dict.depth_range[0][0]."""
    H, W, B, N = 12, 16, 2, 29
    opt = small_opt(nerf=dict(rand_rays=64))
    graph = build(opt, 63)
    pose, intr = ring_cameras(B, H=H, W=W)
    data = data_dict(B, H, W, pose, intr)
    rs = np.random.RandomState(7)
    dmax = T(rs.uniform(2.0, 5.0, size=(N,)).astype(np.float32))
    if sel == "pixels":
        px = T(rs.uniform(0, [W - 1, H - 1], size=(N, 2)).astype(np.float32))
        kw = dict(pixels=px.to(dev()))
        center, ray = O.rays_at_pixels(pose[:1], intr[:1], px[None])
    else:
        idx = T(rs.permutation(H * W)[:N])
        kw = dict(ray_idx=idx.to(dev()))
        center, ray = O.rays_at_index(pose[:1], intr[:1], H, W, idx)
    with torch.no_grad():
        ret = graph.render_up_to_maxdepth_at_specific_pose_and_rays(opt, data, pose[0].to(dev()), intr[0].to(dev()), H, W, depth_max=dmax.to(dev()),
                                                                   iter=3, mode="train", **kw)
        ref = O.render_to_max(opt, *sds(graph), center, ray, torch.tensor(1.2), dmax[None], mode="train", it=3)
    assert not ret.all_cumulated_fine.requires_grad and ret.all_cumulated_fine.shape == (1, N)
    assert torch.equal(ret.t.cpu(), ref["t"])
    for k in ("all_cumulated", "all_cumulated_fine", "rgb", "depth", "rgb_fine", "depth_fine", "weights_fine"):
        assert max_rel(ret[k], ref[k]) < COARSE_TOL, (k, max_rel(ret[k], ref[k]))
    assert float(ret.all_cumulated_fine.max()) <= 1.0           # depth_cons_loss.py:274 asserts this


def test_public_subapis_on_reference_golden_vectors(golden):
    """NeRF.forward (frequency_nerf.py:172-226), NeRF.positional_encoding (:229-258), FrequencyEmbedder
    (:47-69) and Graph.sample_depth_from_pdf (renderer.py:421-456) fed the inputs of the reference's own
    golden vectors (tests/golden/{mlp,pe,sample_pdf}.npz) through the public methods."""
    # positional encoding + c2f mask
    g = golden("pe")
    x = T(g["in_x"]).to(dev())
    for tag, c2f, prog in (("plain", None, 1.0), ("c2f", [0.4, 0.7], 0.5), ("c2f_lo", [0.4, 0.7], 0.41)):
        opt = small_opt(barf_c2f=c2f)
        graph = build(opt, 21, prog)
        for L in (10, 4):
            enc = graph.nerf.positional_encoding(opt, x, graph.embedder_pts, L)
            np.testing.assert_allclose(enc.cpu().numpy(), g[f"out_{tag}_L{L}"], rtol=2e-5, atol=2e-5)
    # MLP on explicit points
    g = golden("mlp")
    pts, ray = T(g["in_pts"]).to(dev()), T(g["in_ray"]).to(dev())
    for tag, c2f, prog, noise_reg, mode in (("eval", None, None, False, None), ("c2f", [0.4, 0.7], 0.55, False, "train")):
        opt = small_opt(barf_c2f=c2f, nerf=dict(density_noise_reg=noise_reg))
        graph = build(opt, 21, prog)
        with torch.no_grad():
            out = graph.nerf.forward(opt, pts, ray, graph.embedder_pts, graph.embedder_view, mode=mode)
        assert max_rel(out["rgb_samples"].reshape(g[f"out_{tag}_rgb"].shape), g[f"out_{tag}_rgb"]) < 1e-4
        assert max_rel(out["density_samples"].reshape(g[f"out_{tag}_density"].shape), g[f"out_{tag}_density"]) < 1e-4
    # inverse-CDF resampling
    g = golden("sample_pdf")
    w = T(g["in_weights"]).to(dev())
    opt = small_opt()
    graph = build(opt, 3)
    real_rand = torch.rand
    for tag in ("metric_det", "metric_rand", "inverse_rand"):
        rng = [float(v) for v in g[f"in_{tag}_range"]]
        grid = T(g[f"in_{tag}_grid"]) if f"in_{tag}_grid" in g else None
        try:
            if grid is not None:
                torch.rand = lambda *s, **k: grid.clone() if (len(s) == 1 and s[0] == grid.numel()) else real_rand(*s, **k)
            got = graph.sample_depth_from_pdf(opt, w, 8, 6, rng, det=grid is None)
        finally:
            torch.rand = real_rand
        assert got.shape == g[f"out_{tag}"].shape
        np.testing.assert_allclose(got.cpu().numpy(), g[f"out_{tag}"], rtol=0, atol=2e-5)


def test_forward_samples_then_composite(golden):
    """The reference calls forward_samples and composite back to back (renderer.py:304-309); here the
    pair is a view over the fused pass and must give what one render gives; a dictionary the caller built
    itself is composited by the stand-alone kernels (C ABI 6; tests/test_abi6_gpu.py holds it to the golden vectors)."""
    opt = small_opt()
    graph = build(opt, 33)
    rs = np.random.RandomState(9)
    c = T(rs.uniform(-0.3, 0.3, size=(2, 11, 3)).astype(np.float32)).to(dev()) + torch.tensor([0.0, 0.0, -3.0], device=dev())
    r = T(rs.uniform(-0.3, 0.3, size=(2, 11, 3)).astype(np.float32)).to(dev()) + torch.tensor([0.0, 0.0, 1.0], device=dev())
    t = T(np.sort(rs.uniform(1.2, 5.2, size=(2, 11, 8, 1)), axis=2).astype(np.float32)).to(dev())
    with torch.no_grad():
        pred = graph.nerf.forward_samples(opt, c, r, t, graph.embedder_pts, graph.embedder_view, mode="val")
        assert set(k for k in pred if not k.startswith("_")) == {"rgb_samples", "density_samples"}
        out = graph.nerf.composite(opt, r, pred, t)
        ref = O.pass_fixed(opt, sds(graph)[0], c.cpu(), r.cpu(), t.cpu(), mode="val")
    for k in ("rgb", "depth", "opacity", "weights", "all_cumulated", "depth_var", "rgb_samples", "density_samples"):
        assert tuple(out[k].shape) == tuple(ref[k].shape), k
        assert max_rel(out[k], ref[k]) < 1e-4, (k, max_rel(out[k], ref[k]))
    with torch.no_grad():
        alone = graph.nerf.composite(opt, r, dict(rgb_samples=ref["rgb_samples"].to(dev()), density_samples=ref["density_samples"].to(dev())), t)
    for k in ("rgb", "depth", "opacity", "weights", "all_cumulated", "depth_var"):       # (rgb_var = (sum_ch rgb)(1 - opacity) is rounding noise around 0 here)
        assert max_rel(alone[k], ref[k]) < 2e-5, (k, max_rel(alone[k], ref[k]))
    assert float((alone["rgb_var"].cpu() - ref["rgb_var"]).abs().max()) < 2e-5


def test_caller_replay_grad_mode_toggles():
    """one SPARF iteration's render calls in the trainers' order with their grad-mode
    toggles -- set_grad_enabled(True) training renders, render_up_to_maxdepth under no_grad in the
    middle (depth_cons_loss.py:266), a validation render after set_grad_enabled(False)
    (base.py:181-195 / eval.py:31), then test-time optimisation re-enabling grad
    (joint_pose_nerf_trainer.py:381).  Nothing saved when grad is off, gradients exact when it is on."""
    H, W, B = 12, 16, 3
    opt = small_opt(nerf=dict(rand_rays=48, sample_stratified=False), barf_c2f=[0.1, 0.5])
    graph = build(opt, 71, progress=0.3)
    pose, intr = ring_cameras(B, H=H, W=W)
    data = data_dict(B, H, W, pose, intr)
    rs = np.random.RandomState(8)
    px = T(rs.uniform(0, [W - 1, H - 1], size=(21, 2)).astype(np.float32))
    dmax = T(rs.uniform(2.0, 5.0, size=(21,)).astype(np.float32))
    idx = T(rs.permutation(H * W)[:16])
    prev = torch.is_grad_enabled()
    try:
        torch.set_grad_enabled(True)                                  # set_train_mode
        pg = pose.to(dev()).requires_grad_(True)
        data.pose = pg
        ret = graph.render_image_at_specific_rays(opt, data, iter=10, ray_idx=idx.to(dev()), mode="train")
        r_self = graph.render_image_at_specific_pose_and_rays(opt, data, pg[0], intr[0].to(dev()), H, W, iter=10, pixels=px.to(dev()), mode="train")
        with torch.no_grad():
            r_max = graph.render_up_to_maxdepth_at_specific_pose_and_rays(opt, data, pg[1].detach(), intr[1].to(dev()), H, W, depth_max=dmax.to(dev()),
                                                                         iter=10, pixels=px.to(dev()), mode="train")
        assert not r_max.all_cumulated_fine.requires_grad and ret.rgb.requires_grad and r_self.depth_fine.requires_grad
        vis = r_max.all_cumulated_fine.reshape(-1)
        loss = ret.rgb.mean() + ret.rgb_fine.mean() + (vis * r_self.depth_fine.reshape(-1)).mean()
        loss.backward()
        g_hip = graph.nerf_fine.mlp_feat[2].weight.grad.clone()
        gp_hip = pg.grad.clone()
        torch.set_grad_enabled(False)                                 # set_eval_mode / eval.py:31
        val = graph.forward(opt, data, iter=None, mode="val")
        assert not val.rgb.requires_grad and val.rgb_fine.shape == (B, H * W, 3)
        torch.set_grad_enabled(True)                                  # test-time photometric optimisation
        pt = pose[:1].to(dev()).requires_grad_(True)
        r_opt = graph.render(opt, pt, H=H, W=W, intr=intr[:1].to(dev()), ray_idx=idx.to(dev()), depth_range=[1.2, 5.2], iter=None, mode="test-optim")
        r_opt.rgb_fine.mean().backward()
        assert pt.grad is not None and float(pt.grad.abs().max()) > 0
    finally:
        torch.set_grad_enabled(prev)
    # the same iteration through the oracle
    sd_c, sd_f = sds(graph)
    sd_f = {k: v.clone().requires_grad_(k != "progress") for k, v in sd_f.items()}
    po = pose.clone().requires_grad_(True)
    rng = [torch.tensor(1.2), torch.tensor(5.2)]
    c1, r1 = O.rays_at_index(po, intr, H, W, idx)
    o1 = O.render(opt, sd_c, sd_f, c1, r1, rng, mode="train", it=10)
    c2, r2 = O.rays_at_pixels(po[:1], intr[:1], px[None])
    o2 = O.render(opt, sd_c, sd_f, c2, r2, rng, mode="train", it=10)
    with torch.no_grad():
        c3, r3 = O.rays_at_pixels(po[1:2], intr[1:2], px[None])
        o3 = O.render_to_max(opt, sd_c, sd_f, c3, r3, torch.tensor(1.2), dmax[None], mode="train", it=10)
    lref = o1["rgb"].mean() + o1["rgb_fine"].mean() + (o3["all_cumulated_fine"].reshape(-1) * o2["depth_fine"].reshape(-1)).mean()
    lref.backward()
    assert abs(float(loss.detach()) - float(lref.detach())) < 1e-4 * abs(float(lref.detach()))
    assert max_rel(g_hip, sd_f["mlp_feat.2.weight"].grad) < 5e-3
    assert max_rel(gp_hip, po.grad) < 2e-2


def test_render_batch_matches_oracle():
    """SURVEY 8f next-2 against the ORACLE: three requests (ray-index render on all views, pixel render
    on one pose, render_to_max under no_grad) through one render_batch call."""
    H, W, B = 10, 12, 3
    opt = small_opt(nerf=dict(rand_rays=32))
    graph = build(opt, 13)
    pose, intr = ring_cameras(B, H=H, W=W)
    rs = np.random.RandomState(2)
    px = T(rs.uniform(0, [W - 1, H - 1], size=(19, 2)).astype(np.float32))
    idx = T(rs.randint(0, H * W, size=(23,)))
    dmax = T(rs.uniform(2.0, 5.0, size=(1, 11)).astype(np.float32))
    pg = pose.to(dev()).requires_grad_(True)
    K = intr.to(dev())
    reqs = [dict(pose=pg, H=H, W=W, intr=K, ray_idx=idx.to(dev()), depth_range=[1.5, 4.0], mode="val"),
            dict(pose=pg[:1], H=H, W=W, intr=K[:1], pixels=px.to(dev()), depth_range=[1.2, 5.2], mode="val"),
            dict(pose=pg[2:], H=H, W=W, intr=K[2:], ray_idx=idx[:11].to(dev()), depth_min=1.2, depth_max=dmax.to(dev()), mode="val", no_grad=True)]
    rets = graph.render_batch(opt, reqs, iter=None)
    loss = rets[0].rgb_fine.sum() + 2 * rets[1].depth_fine.sum() + rets[0].rgb.sum()
    loss.backward()
    sd_c, sd_f = sds(graph)
    sd_f = {k: v.clone().requires_grad_(k != "progress") for k, v in sd_f.items()}
    po = pose.clone().requires_grad_(True)
    c0, r0 = O.rays_at_index(po, intr, H, W, idx)
    o0 = O.render(opt, sd_c, sd_f, c0, r0, [1.5, 4.0], mode="val", it=None)
    c1, r1 = O.rays_at_pixels(po[:1], intr[:1], px[None])
    o1 = O.render(opt, sd_c, sd_f, c1, r1, [1.2, 5.2], mode="val", it=None)
    with torch.no_grad():
        c2, r2 = O.rays_at_index(po[2:], intr[2:], H, W, idx[:11])
        o2 = O.render_to_max(opt, sd_c, sd_f, c2, r2, 1.2, dmax, mode="val", it=None)
    lref = o0["rgb_fine"].sum() + 2 * o1["depth_fine"].sum() + o0["rgb"].sum()
    lref.backward()
    check_render(rets[0], o0)
    check_render(rets[1], o1)
    assert not rets[2].all_cumulated_fine.requires_grad
    for k in ("all_cumulated", "all_cumulated_fine", "depth", "rgb_fine"):
        assert max_rel(rets[2][k], o2[k]) < COARSE_TOL, k
    assert abs(float(loss.detach()) - float(lref.detach())) < 2e-3 * abs(float(lref.detach()))
    assert max_rel(graph.nerf_fine.mlp_feat[5].weight.grad, sd_f["mlp_feat.5.weight"].grad) < 0.3      # no c2f: conditioning-limited (test_graph_gpu.py)


def test_render_batch_equals_separate_calls_and_routes_gradients():
    """The shared-buffer / segment-table path (include/sparf_hip.h sparf_segment_t) against the SAME requests issued
    one by one: deterministic mode, so outputs are bit-identical (rays are independent) and parameter / pose
    gradients agree to summation order; a train-mode request next to a val-mode one only adds noise to its own rows."""
    H, W, B = 10, 12, 3
    opt = small_opt(nerf=dict(rand_rays=32, density_noise_reg=True))
    pose, intr = ring_cameras(B, H=H, W=W)
    rs = np.random.RandomState(5)
    px = T(rs.uniform(0, [W - 1, H - 1], size=(21, 2)).astype(np.float32)).to(dev())
    idx = T(rs.randint(0, H * W, size=(17,))).to(dev())
    K = intr.to(dev())

    def requests(pg):
        return [dict(pose=pg, H=H, W=W, intr=K, ray_idx=idx, depth_range=[1.5, 4.0], mode="val"),
                dict(pose=pg[1:2], H=H, W=W, intr=K[1:2], pixels=px, depth_range=[1.2, 5.2], mode="val"),
                dict(pose=pg[0:1].detach(), H=H, W=W, intr=K[0:1], pixels=px[:9], depth_range=[1.2, 5.2], mode="val")]

    def loss_of(rets):
        return rets[0].rgb_fine.sum() + 2 * rets[1].depth_fine.sum() + rets[0].rgb.sum() + 0.5 * rets[2].opacity.sum() + (rets[1].weights_fine ** 2).sum()

    results = {}
    for how in ("batch", "separate"):
        graph = build(opt, 13)
        pg = pose.to(dev()).requires_grad_(True)
        reqs = requests(pg)
        if how == "batch":
            rets = graph.render_batch(opt, reqs, iter=None)
        else:
            rets = [graph.render(opt, q["pose"], H=H, W=W, intr=q["intr"], pixels=q.get("pixels"), ray_idx=q.get("ray_idx"),
                                 depth_range=q["depth_range"], iter=None, mode="val") for q in reqs]
        loss_of(rets).backward()
        results[how] = (rets, pg.grad.clone(), {k: p.grad.clone() for k, p in graph.named_parameters() if p.grad is not None})
    rb, gpb, gb = results["batch"]
    rsep, gps, gs = results["separate"]
    for a, b in zip(rb, rsep):
        for k in ("rgb", "depth", "opacity", "weights", "rgb_fine", "depth_fine", "weights_fine", "t", "t_fine", "origins", "viewdirs", "all_cumulated_fine"):
            assert tuple(a[k].shape) == tuple(b[k].shape), k
            assert torch.equal(a[k], b[k]), (k, max_rel(a[k], b[k].cpu()))
    assert max_rel(gpb, gps.cpu()) < 1e-4
    assert set(gb) == set(gs)
    for k in gs:
        assert max_rel(gb[k], gs[k].cpu()) < 2e-3, (k, max_rel(gb[k], gs[k].cpu()))      # bf16-free fp32 default mode: split-K partition differs with the row count

    # train-mode request next to a val-mode one: the val rows see no noise
    graph = build(opt, 13)
    pg = pose.to(dev())
    with torch.no_grad():
        q_val = dict(pose=pg[1:2], H=H, W=W, intr=K[1:2], pixels=px, depth_range=[1.2, 5.2], mode="val")
        q_train = dict(pose=pg, H=H, W=W, intr=K, ray_idx=idx, depth_range=[1.5, 4.0], mode="train")
        mixed = graph.render_batch(opt, [q_train, q_val], iter=None)
        alone = graph.render(opt, q_val["pose"], H=H, W=W, intr=q_val["intr"], pixels=px, depth_range=[1.2, 5.2], iter=None, mode="val")
        quiet = graph.render(opt, pg, H=H, W=W, intr=K, ray_idx=idx, depth_range=[1.5, 4.0], iter=None, mode="val")
    assert torch.equal(mixed[1].rgb, alone.rgb) and torch.equal(mixed[1].rgb_fine, alone.rgb_fine)
    assert not torch.equal(mixed[0].density_samples, quiet.density_samples)         # the train rows did get their noise


def test_render_batch_more_requests_than_one_segment_table():
    """18 requests in one call: the segment table of a pass holds SPARF_MAX_SEGMENTS = 16, so the group runs as two
    consecutive passes -- same results as 18 separate calls, gradients included."""
    H, W, B = 10, 12, 2
    opt = small_opt(nerf=dict(rand_rays=32))
    pose, intr = ring_cameras(B, H=H, W=W)
    K = intr.to(dev())
    rs = np.random.RandomState(7)
    idxs = [T(rs.randint(0, H * W, size=(3 + i % 4,))).to(dev()) for i in range(18)]
    out = {}
    for how in ("batch", "separate"):
        graph = build(opt, 17)
        pg = pose.to(dev()).requires_grad_(True)
        reqs = [dict(pose=pg[i % B:i % B + 1], H=H, W=W, intr=K[i % B:i % B + 1], ray_idx=ix, depth_range=[1.2, 5.2], mode="val") for i, ix in enumerate(idxs)]
        if how == "batch":
            rets = graph.render_batch(opt, reqs, iter=None)
        else:
            rets = [graph.render(opt, q["pose"], H=H, W=W, intr=q["intr"], ray_idx=q["ray_idx"], depth_range=q["depth_range"], iter=None, mode="val") for q in reqs]
        sum((i + 1) * r.rgb_fine.sum() + r.depth.sum() for i, r in enumerate(rets)).backward()
        out[how] = (rets, pg.grad.clone(), graph.nerf_fine.mlp_feat[2].weight.grad.clone())
    for a, b in zip(out["batch"][0], out["separate"][0]):
        assert torch.equal(a.rgb_fine, b.rgb_fine) and torch.equal(a.depth, b.depth) and torch.equal(a.t_fine, b.t_fine)
    assert max_rel(out["batch"][1], out["separate"][1].cpu()) < 1e-4
    assert max_rel(out["batch"][2], out["separate"][2].cpu()) < 2e-3


def test_progress_write_through_data_takes_effect_immediately():
    """the trainer moves BARF c2f by progress.data.fill_(x) with no weight update in
    between (gradient accumulation, evaluation at several progress values).  Every render must use the
    CURRENT value: compared with a fresh Graph loaded at that progress."""
    H, W, B = 8, 10, 2
    opt = small_opt(barf_c2f=[0.1, 0.5], nerf=dict(rand_rays=32))
    pose, intr = ring_cameras(B, H=H, W=W)
    idx = torch.arange(20, device=dev())
    graph = build(opt, 81, progress=0.15)

    def render(g):
        with torch.no_grad():
            return g.render(opt, pose.to(dev()), H=H, W=W, intr=intr.to(dev()), ray_idx=idx, depth_range=[1.2, 5.2], iter=None, mode="val")
    first = render(graph)
    for x in (0.3, 0.45, 1.0):
        graph.nerf.progress.data.fill_(x)
        graph.nerf_fine.progress.data.fill_(x)
        got = render(graph)
        fresh = render(build(opt, 81, progress=x))
        assert torch.equal(got.rgb, fresh.rgb) and torch.equal(got.rgb_fine, fresh.rgb_fine), x
        assert not torch.equal(got.rgb, first.rgb)


def test_forward_with_img_idx_and_mask_img():
    """Graph.forward(img_idx=...) (renderer.py:111-121: rand_rays // n_img random rays of the selected images,
    `ray_idx` and `idx_img_rendered` attached) and opt.mask_img = True (white background added like
    setbg_opaque, frequency_nerf.py:337-338) against the oracle on the rays forward() drew."""
    H, W, B = 12, 16, 3
    opt = small_opt(mask_img=True, nerf=dict(rand_rays=30, sample_stratified=False))
    graph = build(opt, 91)
    pose, intr = ring_cameras(B, H=H, W=W)
    data = data_dict(B, H, W, pose, intr)
    with torch.no_grad():
        ret = graph.forward(opt, data, iter=3, img_idx=[0, 2], mode="train")
        assert ret.idx_img_rendered == [0, 2] and ret.ray_idx.shape == (15,) and ret.rgb.shape == (2, 15, 3)
        one = graph.forward(opt, data, iter=3, img_idx=1, mode="train")
        assert one.rgb.shape == (1, 30, 3) and one.ray_idx.shape == (30,)
        sel = torch.tensor([0, 2])
        center, ray = O.rays_at_index(pose[sel], intr[sel], H, W, ret.ray_idx.cpu())
        ref = O.render(opt, *sds(graph), center, ray, [torch.tensor(1.2), torch.tensor(5.2)], mode="train", it=3)
    check_render(ret, ref)
    assert float((ret.rgb.cpu() - ref["rgb"]).abs().max()) < 5e-4           # includes the 1 - opacity background term
