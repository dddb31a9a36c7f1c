"""Generate golden vectors by running the REFERENCE renderer itself.

Runs only in the build container (needs /root/reference, which does not exist
on the GPU box).  Usage:  python tests/golden/make_golden.py
Writes tests/golden/*.npz.  Inputs are rebuilt from seeds by recipe.py; every
random tensor the reference draws (torch.rand jitter / fine grid,
torch.randn_like sigma noise) is captured and stored so the oracle and the HIP
path can be fed the identical values.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat"), "/root/reference"]

from source.models.renderer import Graph            # noqa: E402  (the reference)
from source.models.frequency_nerf import NeRF, FrequencyEmbedder  # noqa: E402
from tests.golden.recipe import small_opt, make_state_dict, ring_cameras  # noqa: E402

torch.set_num_threads(4)


class Capture:
    """Record what torch.rand / torch.randn_like return inside the reference."""

    def __init__(self):
        self.rand, self.randn = [], []

    def __enter__(self):
        self._rand, self._randn_like = torch.rand, torch.randn_like

        def rand(*a, **k):
            out = self._rand(*a, **k)
            self.rand.append(out.detach().cpu().clone())
            return out

        def randn_like(*a, **k):
            out = self._randn_like(*a, **k)
            self.randn.append(out.detach().cpu().clone())
            return out

        torch.rand, torch.randn_like = rand, randn_like
        return self

    def __exit__(self, *exc):
        torch.rand, torch.randn_like = self._rand, self._randn_like


def npy(d):
    out = {}
    for k, v in d.items():
        if v is None:
            continue
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    return out


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    arrays = npy(arrays)
    np.savez_compressed(path, **arrays)
    print("wrote", path, {k: tuple(v.shape) for k, v in arrays.items()})


def build_graph(opt, seed, progress=None):
    g = Graph(opt, torch.device("cpu"))
    g.nerf.load_state_dict(make_state_dict(opt, seed, progress))
    if opt.nerf.fine_sampling:
        g.nerf_fine.load_state_dict(make_state_dict(opt, seed + 1, progress))
    return g


# --------------------------------------------------------------------------- PE
def gen_pe():
    rs = np.random.RandomState(11)
    x = torch.from_numpy(rs.uniform(-2.5, 2.5, size=(2, 3, 4, 3)).astype(np.float32))
    out = {"in_x": x}
    for tag, c2f, prog in (("plain", None, None), ("c2f", [0.4, 0.7], 0.5), ("c2f_lo", [0.4, 0.7], 0.41)):
        opt = small_opt(barf_c2f=c2f)
        net = NeRF(opt)
        if prog is not None:
            net.progress.data.fill_(prog)
        emb = FrequencyEmbedder(opt)
        for L in (10, 4):
            out[f"out_{tag}_L{L}"] = net.positional_encoding(opt, x, embedder_fn=emb, L=L)
    save("pe", **out)


# -------------------------------------------------------------------------- MLP
def gen_mlp():
    rs = np.random.RandomState(12)
    pts = torch.from_numpy(rs.uniform(-1.5, 1.5, size=(1, 3, 5, 3)).astype(np.float32))
    ray = torch.from_numpy(rs.uniform(-1, 1, size=(1, 3, 3)).astype(np.float32))
    out = {"in_pts": pts, "in_ray": ray}
    for tag, c2f, prog, noise_reg, mode in (("eval", None, None, False, None),
                                            ("train_noise", None, None, True, "train"),
                                            ("c2f", [0.4, 0.7], 0.55, False, "train")):
        opt = small_opt(barf_c2f=c2f, nerf=dict(density_noise_reg=noise_reg))
        g = build_graph(opt, 21, prog)
        with Capture() as cap:
            pred = g.nerf.forward(opt, pts, ray, g.embedder_pts, g.embedder_view, mode=mode)
        out[f"out_{tag}_rgb"] = pred["rgb_samples"]
        out[f"out_{tag}_density"] = pred["density_samples"]
        if cap.randn:
            out[f"in_{tag}_noise"] = cap.randn[0]
    save("mlp", **out)


# -------------------------------------------------------------------- composite
def gen_composite():
    rs = np.random.RandomState(13)
    B, R, N = 2, 3, 7
    rgb_s = torch.from_numpy(rs.uniform(0, 1, size=(B, R, N, 3)).astype(np.float32))
    dens = torch.from_numpy(rs.gamma(1.0, 2.0, size=(B, R, N)).astype(np.float32))
    t = torch.from_numpy(np.sort(rs.uniform(1.2, 5.2, size=(B, R, N, 1)), axis=2).astype(np.float32))
    ray = torch.from_numpy(rs.uniform(-1, 1, size=(B, R, 3)).astype(np.float32))
    out = {"in_rgb_s": rgb_s, "in_density": dens, "in_t": t, "in_ray": ray}
    for tag, bg in (("plain", False), ("bg", True)):
        opt = small_opt(nerf=dict(setbg_opaque=bg))
        net = NeRF(opt)
        pred = net.composite(opt, ray, dict(rgb_samples=rgb_s.clone(), density_samples=dens.clone()), t)
        for k in ("rgb", "rgb_var", "depth", "depth_var", "opacity", "weights", "all_cumulated"):
            out[f"out_{tag}_{k}"] = pred[k]
    save("composite", **out)


# ------------------------------------------------------------------- sample_pdf
def gen_sample_pdf():
    rs = np.random.RandomState(14)
    w = torch.from_numpy(rs.gamma(0.5, 1.0, size=(2, 3, 8)).astype(np.float32))
    w = w / w.sum(-1, keepdim=True) * torch.from_numpy(rs.uniform(0.3, 1.0, size=(2, 3, 1)).astype(np.float32))
    w[0, 1, 5:] = 0.0          # a saturated ray: flat CDF tail
    out = {"in_weights": w}
    opt = small_opt()
    g = build_graph(opt, 3)
    for tag, rng, det in (("metric_det", [1.2, 5.2], True), ("metric_rand", [1.2, 5.2], False),
                          ("inverse_rand", [1, 0], False)):
        with Capture() as cap:
            tf = g.sample_depth_from_pdf(opt, weights=w, n_samples_coarse=8, n_samples_fine=6,
                                         depth_range=rng, det=det)
        out[f"out_{tag}"] = tf
        out[f"in_{tag}_range"] = np.array(rng, dtype=np.float32)
        if cap.rand:
            out[f"in_{tag}_grid"] = cap.rand[0]
    save("sample_pdf", **out)


# ----------------------------------------------------------------------- render
RENDER_KEYS = ("origins", "viewdirs", "rgb_samples", "density_samples", "t", "rgb", "rgb_var", "depth",
               "depth_var", "opacity", "weights", "all_cumulated")


def dump_render(out, tag, ret, cap, n_noise_expected):
    for k in RENDER_KEYS:
        for suf in ("", "_fine"):
            if k + suf in ret:
                out[f"out_{tag}__{k}{suf}"] = ret[k + suf]
    # torch.rand order inside Graph.render: jitter (4-D), then the fine grid (1-D)
    for r in cap.rand:
        if r.dim() == 4:
            out[f"in_{tag}__jitter"] = r
        elif r.dim() == 1:
            out[f"in_{tag}__grid"] = r
    assert len(cap.randn) == n_noise_expected, (tag, len(cap.randn))
    for i, n in enumerate(cap.randn):
        out[f"in_{tag}__noise{'_fine' if i else ''}"] = n


def gen_render():
    H, W, B = 6, 8, 2
    pose, intr = ring_cameras(B, seed=0, H=H, W=W)
    out = {"in_pose": pose, "in_intr": intr, "in_HW": np.array([H, W])}
    rs = np.random.RandomState(15)
    idx_shared = torch.from_numpy(rs.permutation(H * W)[:5].astype(np.int64))
    idx_per = torch.from_numpy(np.stack([rs.permutation(H * W)[:5] for _ in range(B)]).astype(np.int64))
    pix = torch.from_numpy(rs.uniform(0, [W, H], size=(B, 5, 2)).astype(np.float32))
    out.update(in_idx_shared=idx_shared, in_idx_per=idx_per, in_pixels=pix)

    cases = [
        # tag, opt overrides, kwargs for render, n sigma-noise draws
        ("metric_train", dict(nerf=dict(density_noise_reg=True)), dict(ray_idx=idx_shared, depth_range=[1.2, 5.2], mode="train", iter=100), 2),
        ("metric_train_peridx", dict(), dict(ray_idx=idx_per, depth_range=[1.2, 5.2], mode="train", iter=100), 0),
        ("inverse_pixels", dict(nerf=dict(depth=dict(param="inverse", range=[1, 0]))), dict(pixels=pix, depth_range=[1, 0], mode="train", iter=100), 0),
        ("metric_val", dict(nerf=dict(density_noise_reg=True)), dict(ray_idx=idx_shared, depth_range=[1.2, 5.2], mode="val", iter=None), 0),
        ("gate_skip", dict(nerf=dict(ratio_start_fine_sampling_at_x=0.5), max_iter=1000), dict(ray_idx=idx_shared, depth_range=[1.2, 5.2], mode="train", iter=10), 0),
        ("c2f_bg", dict(barf_c2f=[0.4, 0.7], nerf=dict(setbg_opaque=True)), dict(pixels=pix, depth_range=[1.2, 5.2], mode="train", iter=100), 0),
    ]
    for tag, over, kw, n_noise in cases:
        opt = small_opt(**over)
        g = build_graph(opt, 31, progress=0.52 if opt.barf_c2f is not None else None)
        with Capture() as cap:
            ret = g.render(opt, pose, H=H, W=W, intr=intr, **kw)
        dump_render(out, tag, ret, cap, n_noise)
    save("render", **out)


def gen_render_to_max():
    H, W, B = 6, 8, 2
    pose, intr = ring_cameras(B, seed=1, H=H, W=W)
    rs = np.random.RandomState(16)
    pix = torch.from_numpy(rs.uniform(0, [W, H], size=(B, 4, 2)).astype(np.float32))
    dmax = torch.from_numpy(rs.uniform(2.0, 5.0, size=(B, 4)).astype(np.float32))
    opt = small_opt()
    g = build_graph(opt, 41)
    out = {"in_pose": pose, "in_intr": intr, "in_pixels": pix, "in_depth_max": dmax, "in_depth_min": np.float32(1.2)}
    with Capture() as cap:
        ret = g.render_to_max(opt, pose, H=H, W=W, intr=intr, pixels=pix, depth_max=dmax, depth_min=1.2, iter=5, mode="train")
    assert not cap.rand and not cap.randn
    for k in RENDER_KEYS:
        for suf in ("", "_fine"):
            if k + suf in ret:
                out[f"out_{k}{suf}"] = ret[k + suf]
    save("render_to_max", **out)


# ------------------------------------------------------------------------ grads
def grad_signature(t):
    """Compact fingerprint of a gradient tensor: [sum, sum|.|, sum of squares]
    followed by a strided sample of up to 64 entries."""
    f = t.detach().reshape(-1).double()
    stride = max(1, f.numel() // 64)
    return torch.cat([torch.stack([f.sum(), f.abs().sum(), (f * f).sum()]), f[::stride][:64]]).numpy()


def gen_grads():
    H, W, B = 6, 8, 2
    pose, intr = ring_cameras(B, seed=2, H=H, W=W)
    rs = np.random.RandomState(17)
    pix = torch.from_numpy(rs.uniform(0, [W, H], size=(B, 5, 2)).astype(np.float32))
    out = {"in_pose": pose, "in_intr": intr, "in_pixels": pix}
    wts = {k: torch.from_numpy(rs.uniform(-1, 1, size=s).astype(np.float32))
           for k, s in (("rgb", (B, 5, 3)), ("depth", (B, 5, 1)), ("opacity", (B, 5, 1)), ("weights", (B, 5, 8, 1)),
                        ("rgb_fine", (B, 5, 3)), ("depth_fine", (B, 5, 1)), ("opacity_fine", (B, 5, 1)),
                        ("weights_fine", (B, 5, 16, 1)))}
    out.update({"in_lw_" + k: v for k, v in wts.items()})
    for tag, over in (("plain", dict(nerf=dict(density_noise_reg=True))),
                      ("c2f_bg", dict(barf_c2f=[0.4, 0.7], nerf=dict(setbg_opaque=True)))):
        opt = small_opt(**over)
        g = build_graph(opt, 51, progress=0.6 if opt.barf_c2f is not None else None)
        p = pose.clone().requires_grad_(True)
        with Capture() as cap:
            ret = g.render(opt, p, H=H, W=W, intr=intr, pixels=pix, depth_range=[1.2, 5.2], mode="train", iter=100)
        ret.origins.retain_grad()         # stage-wise fixtures: the rays / depths the two passes saw and the
        ret.viewdirs.retain_grad()        # gradient that arrives at the rays (test_stagewise_gradients_match_reference)
        loss = sum((ret[k] * w).sum() for k, w in wts.items())
        loss.backward()
        out[f"out_{tag}_loss"] = loss.detach()
        out[f"out_{tag}_dpose"] = p.grad
        for k in ("origins", "viewdirs", "t", "t_fine"):
            out[f"out_{tag}_{k}"] = ret[k]
        out[f"out_{tag}_d_origins"] = ret.origins.grad
        out[f"out_{tag}_d_viewdirs"] = ret.viewdirs.grad
        for r in cap.rand:
            out[f"in_{tag}_jitter" if r.dim() == 4 else f"in_{tag}_grid"] = r
        for i, n in enumerate(cap.randn):
            out[f"in_{tag}_noise{'_fine' if i else ''}"] = n
        for net_name, net in (("nerf", g.nerf), ("nerf_fine", g.nerf_fine)):
            for k, prm in net.named_parameters():
                if k == "progress":
                    assert prm.grad is None          # progress is read through .data
                    continue
                out[f"out_{tag}_grad_{net_name}.{k}"] = grad_signature(prm.grad)
    save("grads", **out)


# ---------------------------------------------------------------- API surface
def gen_api():
    """Signatures of the public methods and the state_dict layout of the reference
    modules, as JSON (the GPU box / CI has no /root/reference to introspect)."""
    import inspect
    import json
    opt = small_opt()
    g = Graph(opt, torch.device("cpu"))
    api = {"Graph": {}, "NeRF": {}, "FrequencyEmbedder": {}}
    for cls_name, obj in (("Graph", Graph), ("NeRF", NeRF), ("FrequencyEmbedder", FrequencyEmbedder)):
        for name, fn in inspect.getmembers(obj, predicate=inspect.isfunction):
            if name.startswith("_") and name not in ("__init__", "__call__"):
                continue
            if name not in obj.__dict__:
                continue
            sig = inspect.signature(fn)
            api[cls_name][name] = [[p.name, None if p.default is inspect._empty else repr(p.default)] for p in sig.parameters.values()]
    api["state_dict"] = {k: list(v.shape) for k, v in g.state_dict().items()}
    with open(os.path.join(HERE, "api.json"), "w") as f:
        json.dump(api, f, indent=1, sort_keys=True)
    print("wrote api.json", {k: len(v) for k, v in api.items()})


if __name__ == "__main__":
    torch.manual_seed(0)
    gen_pe()
    gen_mlp()
    gen_composite()
    gen_sample_pdf()
    gen_render()
    gen_render_to_max()
    gen_grads()
    gen_api()
