"""Pin the oracle: every oracle function vs outputs of the REFERENCE itself
(tests/golden/*.npz, produced by tests/golden/make_golden.py from
/root/reference).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from tests.golden.recipe import small_opt, make_state_dict

T = torch.from_numpy
TOL = dict(rtol=2e-6, atol=2e-6)


def close(a, b, **kw):
    kw = {**TOL, **kw}
    a = a.detach().numpy() if torch.is_tensor(a) else a
    np.testing.assert_allclose(a, b, **kw)


def test_pe(golden):
    g = golden("pe")
    x = T(g["in_x"])
    for tag, c2f, prog in (("plain", None, 1.0), ("c2f", [0.4, 0.7], 0.5), ("c2f_lo", [0.4, 0.7], 0.41)):
        opt = small_opt(barf_c2f=c2f)
        for L in (10, 4):
            close(O.positional_encoding(opt, x, L, torch.tensor(prog)), g[f"out_{tag}_L{L}"])


def test_c2f_mask_probe():
    # progress 0.5 with [0.4,0.7], L=10 -> [1,1,1,0.25,0,...]
    opt = small_opt(barf_c2f=[0.4, 0.7])
    w = O.c2f_mask(opt, 10, torch.tensor(0.5))
    close(w, np.array([1, 1, 1, 0.25] + [0] * 6, dtype=np.float32), atol=1e-6)


def test_mlp(golden):
    g = golden("mlp")
    pts, ray = T(g["in_pts"]), T(g["in_ray"])
    for tag, c2f, prog, noise_reg, mode in (("eval", None, None, False, None),
                                            ("train_noise", None, None, True, "train"),
                                            ("c2f", [0.4, 0.7], 0.55, False, "train")):
        opt = small_opt(barf_c2f=c2f, nerf=dict(density_noise_reg=noise_reg))
        p = make_state_dict(opt, 21, prog)
        noise = T(g[f"in_{tag}_noise"]) if f"in_{tag}_noise" in g else None
        rgb, dens = O.mlp(opt, p, pts, ray, mode=mode, noise=noise)
        close(rgb, g[f"out_{tag}_rgb"])
        close(dens, g[f"out_{tag}_density"], rtol=1e-5)


def test_composite(golden):
    g = golden("composite")
    for tag, bg in (("plain", False), ("bg", True)):
        opt = small_opt(nerf=dict(setbg_opaque=bg))
        out = O.composite(opt, T(g["in_ray"]), T(g["in_rgb_s"]), T(g["in_density"]), T(g["in_t"]))
        for k, v in out.items():
            close(v, g[f"out_{tag}_{k}"])


def test_sample_pdf(golden):
    g = golden("sample_pdf")
    w = T(g["in_weights"])
    for tag in ("metric_det", "metric_rand", "inverse_rand"):
        rng = [float(v) for v in g[f"in_{tag}_range"]]
        grid = T(g[f"in_{tag}_grid"]) if f"in_{tag}_grid" in g else O.det_grid(6)
        close(O.sample_pdf(w, 8, 6, rng, grid), g[f"out_{tag}"])


RENDER_CASES = [
    ("metric_train", dict(nerf=dict(density_noise_reg=True)), "idx_shared", [1.2, 5.2], "train", 100),
    ("metric_train_peridx", dict(), "idx_per", [1.2, 5.2], "train", 100),
    ("inverse_pixels", dict(nerf=dict(depth=dict(param="inverse", range=[1, 0]))), "pixels", [1, 0], "train", 100),
    ("metric_val", dict(nerf=dict(density_noise_reg=True)), "idx_shared", [1.2, 5.2], "val", None),
    ("gate_skip", dict(nerf=dict(ratio_start_fine_sampling_at_x=0.5), max_iter=1000), "idx_shared", [1.2, 5.2], "train", 10),
    ("c2f_bg", dict(barf_c2f=[0.4, 0.7], nerf=dict(setbg_opaque=True)), "pixels", [1.2, 5.2], "train", 100),
]


def rays_for(g, sel):
    pose, intr = T(g["in_pose"]), T(g["in_intr"])
    if sel == "pixels":
        return O.rays_at_pixels(pose, intr, T(g["in_pixels"]))
    H, W = (int(v) for v in g["in_HW"])
    return O.rays_at_index(pose, intr, H, W, T(g["in_" + sel]))


@pytest.mark.parametrize("tag,over,sel,rng,mode,it", RENDER_CASES, ids=[c[0] for c in RENDER_CASES])
def test_render(golden, tag, over, sel, rng, mode, it):
    g = golden("render")
    opt = small_opt(**over)
    prog = 0.52 if opt.barf_c2f is not None else None
    pc, pf = make_state_dict(opt, 31, prog), make_state_dict(opt, 32, prog)
    center, ray = rays_for(g, sel)
    get = lambda k: T(g[f"in_{tag}__{k}"]) if f"in_{tag}__{k}" in g else None
    out = O.render(opt, pc, pf, center, ray, rng, mode=mode, it=it, jitter=get("jitter"), grid=get("grid"),
                   noise_c=get("noise"), noise_f=get("noise_fine"))
    ref_keys = {k[len(f"out_{tag}__"):] for k in g if k.startswith(f"out_{tag}__")}
    assert set(out.keys()) == ref_keys
    for k in sorted(ref_keys):
        # inverse depth puts t up to 1e8: relative tolerance only there
        close(out[k], g[f"out_{tag}__{k}"], rtol=2e-5, atol=2e-5 if "inverse" not in tag else 1e-3)
    if tag == "gate_skip":
        assert "rgb_fine" not in out


def test_render_to_max(golden):
    g = golden("render_to_max")
    opt = small_opt()
    pc, pf = make_state_dict(opt, 41), make_state_dict(opt, 42)
    center, ray = O.rays_at_pixels(T(g["in_pose"]), T(g["in_intr"]), T(g["in_pixels"]))
    out = O.render_to_max(opt, pc, pf, center, ray, float(g["in_depth_min"]), T(g["in_depth_max"]), mode="train", it=5)
    ref_keys = {k[4:] for k in g if k.startswith("out_")}
    assert set(out.keys()) == ref_keys
    for k in sorted(ref_keys):
        close(out[k], g["out_" + k], rtol=2e-5, atol=2e-5)


def grad_signature(t):
    f = t.detach().reshape(-1).double()
    stride = max(1, f.numel() // 64)
    return torch.cat([torch.stack([f.sum(), f.abs().sum(), (f * f).sum()]), f[::stride][:64]]).numpy()


@pytest.mark.parametrize("tag,over", [("plain", dict(nerf=dict(density_noise_reg=True))),
                                      ("c2f_bg", dict(barf_c2f=[0.4, 0.7], nerf=dict(setbg_opaque=True)))])
def test_grads(golden, tag, over):
    g = golden("grads")
    opt = small_opt(**over)
    prog = 0.6 if opt.barf_c2f is not None else None
    pc, pf = make_state_dict(opt, 51, prog), make_state_dict(opt, 52, prog)
    for p in (pc, pf):
        for k, v in p.items():
            if k != "progress":
                v.requires_grad_(True)
    pose = T(g["in_pose"]).clone().requires_grad_(True)
    center, ray = O.rays_at_pixels(pose, T(g["in_intr"]), T(g["in_pixels"]))
    get = lambda k: T(g[f"in_{tag}_{k}"]) if f"in_{tag}_{k}" in g else None
    out = O.render(opt, pc, pf, center, ray, [1.2, 5.2], mode="train", it=100, jitter=get("jitter"),
                   grid=get("grid"), noise_c=get("noise"), noise_f=get("noise_fine"))
    loss = sum((out[k[6:]] * T(v)).sum() for k, v in g.items() if k.startswith("in_lw_"))
    loss.backward()
    close(loss, g[f"out_{tag}_loss"], rtol=1e-5)
    close(pose.grad, g[f"out_{tag}_dpose"], rtol=2e-4, atol=2e-4)
    for net, p in (("nerf", pc), ("nerf_fine", pf)):
        for k, v in p.items():
            if k == "progress":
                continue
            ref = g[f"out_{tag}_grad_{net}.{k}"]
            sig = grad_signature(v.grad)
            scale = max(1.0, float(np.abs(ref[3:]).max()))
            np.testing.assert_allclose(sig[3:], ref[3:], rtol=1e-4, atol=2e-5 * scale)
            np.testing.assert_allclose(sig[:3], ref[:3], rtol=2e-4, atol=1e-3)


@pytest.mark.parametrize("tag,over", [("plain", dict(nerf=dict(density_noise_reg=True))),
                                      ("c2f_bg", dict(barf_c2f=[0.4, 0.7], nerf=dict(setbg_opaque=True)))])
@pytest.mark.parametrize("referee", [False, True], ids=["fp32", "float64_referee"])
def test_stagewise_grads(golden, tag, over, referee):
    """The two passes on the reference's OWN rays and depths (origins, viewdirs, t, t_fine of its
    render) through oracle.pass_fixed: the gradient reaching the rays and every parameter gradient
    equal the reference's end-to-end ones (t_fine carries no gradient).  Also pins the float64
    referee mode used by the benchmark-scale parity tests to the reference's autograd."""
    g = golden("grads")
    opt = small_opt(**over)
    prog = 0.6 if opt.barf_c2f is not None else None
    pc, pf = make_state_dict(opt, 51, prog), make_state_dict(opt, 52, prog)
    cd = torch.float64 if referee else None
    if referee:
        pc, pf = {k: v.double() for k, v in pc.items()}, {k: v.double() for k, v in pf.items()}
    for p in (pc, pf):
        for k, v in p.items():
            if k != "progress":
                v.requires_grad_(True)
    get = lambda k: T(g[f"in_{tag}_{k}"]) if f"in_{tag}_{k}" in g else None
    c, r = T(g[f"out_{tag}_origins"]).requires_grad_(True), T(g[f"out_{tag}_viewdirs"]).requires_grad_(True)
    oc = O.pass_fixed(opt, pc, c, r, T(g[f"out_{tag}_t"]), mode="train", noise=get("noise"), compute_dtype=cd)
    of = O.pass_fixed(opt, pf, c, r, T(g[f"out_{tag}_t_fine"]), mode="train", noise=get("noise_fine"), fine=True, compute_dtype=cd)
    out = dict(oc)
    out.update({k + "_fine": v for k, v in of.items()})
    loss = sum((out[k[6:]] * T(v).to(out[k[6:]].dtype)).sum() for k, v in g.items() if k.startswith("in_lw_"))
    loss.backward()
    close(loss.float(), g[f"out_{tag}_loss"], rtol=1e-5)
    # in the reference ray = X_world - center (camera.py:411-412), so what its autograd leaves on `origins` is
    # the direct gradient minus the one routed through the ray
    close(c.grad - r.grad, g[f"out_{tag}_d_origins"], rtol=2e-4, atol=2e-4 * float(np.abs(g[f"out_{tag}_d_origins"]).max()))
    close(r.grad, g[f"out_{tag}_d_viewdirs"], rtol=2e-4, atol=2e-4 * float(np.abs(g[f"out_{tag}_d_viewdirs"]).max()))
    for net, p in (("nerf", pc), ("nerf_fine", pf)):
        for k, v in p.items():
            if k == "progress":
                continue
            ref = g[f"out_{tag}_grad_{net}.{k}"]
            sig = grad_signature(v.grad)
            scale = max(1e-6, float(np.abs(ref[3:]).max()))
            np.testing.assert_allclose(sig[3:], ref[3:], rtol=1e-4, atol=2e-5 * scale)


def test_init_matches_reference_statistics():
    """init_params follows tensorflow_init_weights' bounds (frequency_nerf.py:136-147)."""
    import math
    opt = small_opt()
    p = O.init_params(opt, seed=0)
    w = p["mlp_feat.1.weight"]
    a = math.sqrt(2) * math.sqrt(6 / 512)
    assert w.abs().max() <= a and w.abs().max() > 0.98 * a
    w7 = p["mlp_feat.7.weight"]
    assert w7.shape == (257, 256) and w7[0].abs().max() <= math.sqrt(6 / 257)
    assert p["mlp_rgb.1.weight"].abs().max() <= math.sqrt(6 / 131)
    assert all(float(v.abs().sum()) == 0 for k, v in p.items() if k.endswith("bias"))
    assert float(p["progress"]) == 1.0 and float(O.init_params(small_opt(barf_c2f=[0.1, 0.5]))["progress"]) == 0.0
    n = sum(v.numel() for v in p.values())
    assert n == 530053
