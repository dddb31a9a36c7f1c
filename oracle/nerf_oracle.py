"""CPU oracle for the SPARF renderer hot path.  *** TEST INFRASTRUCTURE ONLY ***

This file is a plain-PyTorch (fp32, or fp64 when fed fp64 tensors) restatement
of the reference algorithm in

    /root/reference/source/models/renderer.py        (Graph)
    /root/reference/source/models/frequency_nerf.py  (FrequencyEmbedder, NeRF)
    /root/reference/source/utils/camera.py:347-437   (ray generation, points)

It exists to CHECK the HIP path.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it; nothing under `sparf_amd/`
does, and the product path raises if the HIP library is missing rather than
falling back to this code.

Parity status: PINNED.  The reference has no tests / golden vectors of its own
(SURVEY.md §4), but it is pure Python and imports in the build container, so
`tests/golden/make_golden.py` runs the *reference modules themselves* on seeded
inputs and commits their outputs under `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every function below against them.

Design differences from the reference (behaviour identical, structure not):
  * functional: parameters are a flat dict keyed like the reference
    state_dict ("mlp_feat.3.weight", "mlp_rgb.0.bias", "progress");
  * all randomness is an explicit input (`jitter`, `grid`, `noise`) because a
    GPU RNG can never bit-match torch's; the reference draws them with
    torch.rand / torch.randn_like at renderer.py:406, :439, frequency_nerf.py:192.
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# architecture bookkeeping  (frequency_nerf.py:87-124)
# ----------------------------------------------------------------------------


def input_dims(opt):
    """3-D and view input widths: raw coords (optional) + 6L encodings."""
    pe = opt.arch.posenc
    d3 = (3 if pe.add_raw_3D_points else 0) + (6 * pe.L_3D if pe.L_3D > 0 else 0)
    dv = 0
    if opt.nerf.view_dep:
        dv = (3 if pe.add_raw_rays else 0) + (6 * pe.L_view if pe.L_view > 0 else 0)
    return d3, dv


def layer_shapes(opt, fine=False):
    """[(state_dict prefix, out_features, in_features)] in module order.

    frequency_nerf.py:102-123: first layer takes the encoded point, layers in
    `arch.skip` take [h, x0], the last feature layer emits 1 (sigma) + width,
    the colour branch takes [feat, view enc]."""
    d3, dv = input_dims(opt)
    lf = opt.arch.layers_feat
    if fine and opt.arch.get("layers_feat_fine", None) is not None:
        lf = opt.arch.layers_feat_fine
    shapes = []
    n = len(lf) - 1
    for li in range(n):
        k_in, k_out = lf[li], lf[li + 1]
        if li == 0:
            k_in = d3
        if li in opt.arch.skip:
            k_in += d3
        if li == n - 1:
            k_out += 1
        shapes.append((f"mlp_feat.{li}", k_out, k_in))
    lr = opt.arch.layers_rgb
    for li in range(len(lr) - 1):
        k_in, k_out = lr[li], lr[li + 1]
        if li == 0:
            k_in = lf[-1] + dv
        shapes.append((f"mlp_rgb.{li}", k_out, k_in))
    return shapes


def _xavier_uniform_(w, gain, gen):
    # torch.nn.init.xavier_uniform_: U(-a, a), a = gain*sqrt(6/(fan_in+fan_out))
    fan_out, fan_in = w.shape
    a = gain * math.sqrt(6.0 / (fan_in + fan_out))
    w.copy_((torch.rand(w.shape, generator=gen, dtype=torch.float64) * 2 - 1).mul_(a).to(w.dtype))


def init_params(opt, seed=0, fine=False, dtype=torch.float32):
    """'TensorFlow-style' init, frequency_nerf.py:136-147: Xavier-uniform with
    relu gain everywhere except the sigma row of the last feature layer and the
    rgb output layer (gain 1); zero biases; progress = 1 without BARF c2f, else
    0 (frequency_nerf.py:79-85).  The RNG stream is this oracle's own (seeded
    numpy-free torch.Generator), not the reference's global one; tests that need
    identical weights load the same dict into both sides."""
    gen = torch.Generator().manual_seed(seed)
    relu_gain = math.sqrt(2.0)
    shapes = layer_shapes(opt, fine)
    n_feat = sum(1 for s in shapes if s[0].startswith("mlp_feat"))
    n_rgb = len(shapes) - n_feat
    p = {}
    for name, k_out, k_in in shapes:
        w = torch.empty(k_out, k_in, dtype=dtype)
        kind, idx = name.split(".")
        idx = int(idx)
        if kind == "mlp_rgb" and idx == n_rgb - 1:
            _xavier_uniform_(w, 1.0, gen)
        elif kind == "mlp_feat" and idx == n_feat - 1:
            _xavier_uniform_(w[:1], 1.0, gen)       # fan computed on the [1,in] slice, as the reference does
            _xavier_uniform_(w[1:], relu_gain, gen)
        else:
            _xavier_uniform_(w, relu_gain, gen)
        p[name + ".weight"] = w
        p[name + ".bias"] = torch.zeros(k_out, dtype=dtype)
    p["progress"] = torch.tensor(1.0 if opt.barf_c2f is None else 0.0, dtype=dtype)
    return p


# ----------------------------------------------------------------------------
# positional encoding  (frequency_nerf.py:47-69, 229-258)
# ----------------------------------------------------------------------------


def pe_freqs(opt, L, like):
    pe = opt.arch.posenc
    if pe.log_sampling:
        f = 2.0 ** torch.arange(L, dtype=torch.float32)
        if pe.include_pi_in_posenc:
            f = f * math.pi            # float32 multiply, exactly as the reference
    else:
        f = torch.linspace(2.0 ** 0.0, 2.0 ** (L - 1), steps=L) * math.pi
    return f.to(device=like.device, dtype=like.dtype if like.dtype == torch.float64 else torch.float32)


def c2f_mask(opt, L, progress):
    """BARF coarse-to-fine band weights w_k (frequency_nerf.py:248-253);
    None when opt.barf_c2f is None.  `progress` is read as data (no grad)."""
    if opt.barf_c2f is None:
        return None
    start, end = opt.barf_c2f
    prog = progress.detach() if torch.is_tensor(progress) else torch.tensor(float(progress))
    alpha = (prog.to(torch.float32) - start) / (end - start) * L
    k = torch.arange(L, dtype=torch.float32, device=alpha.device)
    return (1 - ((alpha - k).clamp(min=0, max=1) * math.pi).cos()) / 2


def positional_encoding(opt, x, L, progress, compute_dtype=None):
    """[..., C] -> [..., 2*C*L]; per coordinate: L sines then L cosines
    (stack on dim -2 then flatten, frequency_nerf.py:65-68), optionally band
    masked (same w_k for sin and cos of every coordinate, :257).

    compute_dtype (referee mode, see `pass_fixed`): the argument x*freq is formed
    in x's own precision -- the reference's fp32 rounding of it is part of the
    function, worth O(1) rad at |x*freq| ~ 1e7 -- and everything after it runs in
    `compute_dtype`."""
    f = pe_freqs(opt, L, x)
    spec = x[..., None] * f                       # [..., C, L]
    if compute_dtype is not None:
        spec = spec.to(compute_dtype)
    enc = torch.stack([spec.sin(), spec.cos()], dim=-2)   # [..., C, 2, L]
    w = c2f_mask(opt, L, progress)
    if w is not None:
        enc = enc * w.to(enc.dtype)
    return enc.reshape(*x.shape[:-1], -1)


# ----------------------------------------------------------------------------
# MLP  (frequency_nerf.py:149-226)
# ----------------------------------------------------------------------------


def mlp(opt, params, points, ray, mode=None, noise=None, fine=False, compute_dtype=None):
    """points [B,R,N,3], ray [B,R,3] -> rgb_samples [B,R,N,3], density [B,R,N].

    compute_dtype: None = everything in the inputs' dtype (the reference);
    torch.float64 = referee mode (encoding arguments in the inputs' fp32, all
    arithmetic after them in float64; `params` must already be float64).

    `noise` (same shape as density) replaces torch.randn_like at
    frequency_nerf.py:192 and is only applied when the reference would apply
    it (density_noise_reg truthy and mode == 'train')."""
    pe = opt.arch.posenc
    prog = params["progress"]
    if pe.L_3D > 0:
        x0 = positional_encoding(opt, points, pe.L_3D, prog, compute_dtype)
        if pe.add_raw_3D_points:
            x0 = torch.cat([points.to(x0.dtype), x0], dim=-1)
    else:
        x0 = points if compute_dtype is None else points.to(compute_dtype)
    shapes = layer_shapes(opt, fine)
    feat_layers = [s[0] for s in shapes if s[0].startswith("mlp_feat")]
    rgb_layers = [s[0] for s in shapes if s[0].startswith("mlp_rgb")]
    h = x0
    raw = None
    for li, name in enumerate(feat_layers):
        if li in opt.arch.skip:
            h = torch.cat([h, x0], dim=-1)                     # order: [h, x0]  (:164)
        h = F.linear(h, params[name + ".weight"], params[name + ".bias"])
        if li == len(feat_layers) - 1:
            raw, h = h[..., 0], h[..., 1:]
        h = F.relu(h)                                          # also after the last layer (:169)
    if opt.nerf.density_noise_reg and mode == "train":
        assert noise is not None, "oracle needs the sigma noise as an explicit input"
        raw = raw + noise * opt.nerf.density_noise_reg
    density = getattr(F, opt.arch.density_activ)(raw)
    if opt.nerf.view_dep:
        d = F.normalize(ray, dim=-1)[..., None, :].expand_as(points)
        if pe.L_view > 0:
            v = positional_encoding(opt, d, pe.L_view, prog, compute_dtype)
            if pe.add_raw_rays:
                v = torch.cat([d.to(v.dtype), v], dim=-1)
        else:
            v = d if compute_dtype is None else d.to(compute_dtype)
        h = torch.cat([h, v], dim=-1)                          # order: [feat, view]  (:213)
    for li, name in enumerate(rgb_layers):
        h = F.linear(h, params[name + ".weight"], params[name + ".bias"])
        if li != len(rgb_layers) - 1:
            h = F.relu(h)
    return h.sigmoid(), density


def points_from_depth(center, ray, t):
    """camera.py:418-437 with multi_samples=True: p = c + r*t, ray unnormalised."""
    return center[:, :, None] + ray[:, :, None] * t


# ----------------------------------------------------------------------------
# compositing  (frequency_nerf.py:283-343)
# ----------------------------------------------------------------------------


def composite(opt, ray, rgb_s, density, t):
    """t [B,R,N,1].  Returns dict(rgb, rgb_var, depth, depth_var, opacity,
    weights [B,R,N,1], all_cumulated [B,R])."""
    ell = ray.norm(dim=-1, keepdim=True)                        # [B,R,1]
    dt = t[..., 1:, 0] - t[..., :-1, 0]
    dt = torch.cat([dt, torch.full_like(dt[..., :1], 1e10)], dim=2)
    sd = density * (dt * ell)
    alpha = 1 - torch.exp(-sd)
    excl = torch.cat([torch.zeros_like(sd[..., :1]), sd[..., :-1]], dim=2).cumsum(dim=2)
    T = torch.exp(-excl)
    all_cum = T[:, :, -2].clone()
    w = (T * alpha)[..., None]
    depth = (t * w).sum(dim=2)
    depth_var = (w * (t - depth.unsqueeze(-1)) ** 2).sum(dim=2)
    rgb = (rgb_s * w).sum(dim=2)
    rgb_var = ((rgb_s - rgb.unsqueeze(-2)).sum(dim=-1, keepdim=True) * w).sum(dim=2)
    opacity = w.sum(dim=2)
    if opt.nerf.setbg_opaque or opt.mask_img:
        rgb = rgb + (1.0 - opacity)
    return dict(rgb=rgb, rgb_var=rgb_var, depth=depth, depth_var=depth_var,
                opacity=opacity, weights=w, all_cumulated=all_cum)


def pass_fixed(opt, params, center, ray, t, mode=None, noise=None, fine=False, compute_dtype=None):
    """One network over GIVEN rays and depth samples: forward_samples + composite
    (renderer.py:304-309 / :338-343) -> dict of both.  center, ray [B,R,3], t [B,R,N,1].

    compute_dtype=torch.float64 is the REFEREE used by the benchmark-scale parity tests: the
    sample points p = c + r*t and the encoding arguments p*2^k*pi keep the reference's fp32
    rounding (they are inputs of sin/cos with |arg| up to 1e7: their rounding is the function),
    everything downstream -- sin/cos, the ten layers, softplus, the transmittance scan, the
    sums -- runs in float64.  A kernel's distance to this referee is its arithmetic error; the
    fp32 reference's own distance to it is the yardstick."""
    cd = compute_dtype
    pts = points_from_depth(center, ray, t)
    if cd is not None:
        params = {k: v.to(cd) for k, v in params.items()}
        noise = noise.to(cd) if noise is not None else None
    rgb_s, dens = mlp(opt, params, pts, ray, mode, noise, fine=fine, compute_dtype=cd)
    out = dict(rgb_samples=rgb_s, density_samples=dens, t=t)
    out.update(composite(opt, ray if cd is None else ray.to(cd), rgb_s, dens, t if cd is None else t.to(cd)))
    return out


# ----------------------------------------------------------------------------
# depth sampling  (renderer.py:383-456, 595-624)
# ----------------------------------------------------------------------------


def sample_depth(opt, B, R, n, depth_range, mode, jitter=None, device="cpu", dtype=torch.float32):
    """Stratified coarse samples.  jitter [B,R,n,1] in [0,1) replaces
    torch.rand (renderer.py:406); it is used iff sample_stratified and mode not
    in val/eval/test, else the bin mid-point 0.5."""
    dmin, dmax = depth_range
    if opt.nerf.sample_stratified and mode not in ("val", "eval", "test"):
        assert jitter is not None
        u = jitter.to(device=device, dtype=dtype).clone()
    else:
        u = torch.full((B, R, n, 1), 0.5, device=device, dtype=dtype)
    u = u + torch.arange(n, device=device)[None, None, :, None].to(dtype)
    t = u / n * (dmax - dmin) + dmin
    if opt.nerf.depth.param == "inverse":
        t = 1 / (t + 1e-8)
    return t


def sample_depth_to_max(n, depth_min, depth_max, dtype=torch.float32):
    """Per-ray far bound (renderer.py:616-621): t_i = (i+1)/n*(max_r-min)+min."""
    B, R = depth_max.shape
    u = torch.ones(B, R, n, 1, dtype=dtype, device=depth_max.device)
    u = u + torch.arange(n, device=depth_max.device)[None, None, :, None].to(dtype)
    return u / n * (depth_max[..., None, None] - depth_min) + depth_min


def sample_pdf(weights, n_coarse, n_fine, depth_range, grid):
    """Inverse-CDF resampling (renderer.py:427-456).  weights [B,R,Nc];
    grid [Nf+1] is linspace(0,1) when deterministic, else the single shared
    unsorted uniform draw of renderer.py:439.  Bins are a uniform linspace over
    depth_range even for inverse depth."""
    dmin, dmax = depth_range
    pdf = weights / (weights.sum(dim=-1, keepdim=True) + 1e-6)
    cdf = pdf.cumsum(dim=-1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    grid = grid.to(device=weights.device, dtype=weights.dtype)
    u = (0.5 * (grid[:-1] + grid[1:])).expand(*cdf.shape[:-1], n_fine).contiguous()
    idx = torch.searchsorted(cdf, u, right=True)
    bins = torch.linspace(dmin, dmax, n_coarse + 1, device=weights.device, dtype=weights.dtype)
    bins = bins.expand(*cdf.shape[:-1], n_coarse + 1)
    lo_i, hi_i = (idx - 1).clamp(min=0), idx.clamp(max=n_coarse)
    d_lo, d_hi = bins.gather(2, lo_i), bins.gather(2, hi_i)
    c_lo, c_hi = cdf.gather(2, lo_i), cdf.gather(2, hi_i)
    frac = (u - c_lo) / (c_hi - c_lo + 1e-8)
    return (d_lo + frac * (d_hi - d_lo))[..., None]


def det_grid(n_fine, device="cpu", dtype=torch.float32):
    return torch.linspace(0, 1, n_fine + 1, device=device, dtype=dtype)


# ----------------------------------------------------------------------------
# ray generation  (camera.py:296-416)  -- boundary feeder, stays PyTorch in the
# product too; restated here so the oracle is self-contained.
# ----------------------------------------------------------------------------


def _invert_pose(pose):
    R, t = pose[..., :3], pose[..., 3:]
    Rt = R.transpose(-1, -2)
    return torch.cat([Rt, -Rt @ t], dim=-1)


def rays_at_pixels(pose_w2c, intr, xy):
    """xy [B,N,2] pixel coordinates used as given (the `pixels` path adds no
    +0.5, camera.py:400-406).  Returns center, ray [B,N,3] (ray unnormalised:
    R_c2w K^-1 [u,v,1])."""
    hom = torch.cat([xy, torch.ones_like(xy[..., :1])], dim=-1)
    cam = hom @ intr.inverse().transpose(-1, -2)
    c2w = _invert_pose(pose_w2c)

    def to_world(X):
        Xh = torch.cat([X, torch.ones_like(X[..., :1])], dim=-1)
        return Xh @ c2w.transpose(-1, -2)

    center = to_world(torch.zeros_like(cam))
    return center, to_world(cam) - center


def rays_at_index(pose_w2c, intr, H, W, ray_idx):
    """`ray_idx` path: pixel centres at +0.5 (camera.py:365-366); ray_idx is
    [N] shared by all images or [B,N] per image (renderer.py:277-291)."""
    B = pose_w2c.shape[0]
    ray_idx = ray_idx.long()
    if ray_idx.dim() == 1:
        ray_idx = ray_idx[None].expand(B, -1)
    x = (ray_idx % W).to(pose_w2c.dtype) + 0.5
    y = torch.div(ray_idx, W, rounding_mode="floor").to(pose_w2c.dtype) + 0.5
    return rays_at_pixels(pose_w2c, intr, torch.stack([x, y], dim=-1))


# ----------------------------------------------------------------------------
# full render  (renderer.py:250-345, 504-593)
# ----------------------------------------------------------------------------


def _fine_gate_skips(opt, it):
    r = opt.nerf.get("ratio_start_fine_sampling_at_x", None) if hasattr(opt.nerf, "get") else getattr(opt.nerf, "ratio_start_fine_sampling_at_x", None)
    return r is not None and it is not None and it < opt.max_iter * r


def render(opt, params_c, params_f, center, ray, depth_range, mode=None, it=None,
           jitter=None, grid=None, noise_c=None, noise_f=None):
    """Coarse pass, optional fine pass on sorted(coarse U resampled) depths.
    Output keys follow renderer.py:298-344 (fine keys suffixed `_fine`)."""
    B, R = ray.shape[:2]
    Nc = opt.nerf.sample_intvs
    t = sample_depth(opt, B, R, Nc, depth_range, mode, jitter, device=ray.device, dtype=ray.dtype)
    out = dict(origins=center, viewdirs=ray)
    rgb_s, dens = mlp(opt, params_c, points_from_depth(center, ray, t), ray, mode, noise_c)
    out.update(rgb_samples=rgb_s, density_samples=dens, t=t)
    out.update(composite(opt, ray, rgb_s, dens, t))
    if opt.nerf.fine_sampling and not _fine_gate_skips(opt, it):
        Nf = opt.nerf.sample_intvs_fine
        det = mode not in ("train", "test-optim") or (not opt.nerf.sample_stratified)
        with torch.no_grad():
            g = det_grid(Nf, ray.device, ray.dtype) if det else grid
            assert g is not None, "oracle needs the fine-sampling grid as an explicit input"
            t_f = sample_pdf(out["weights"][..., 0], Nc, Nf, depth_range, g)
        t_all = torch.cat([t, t_f], dim=2).sort(dim=2).values
        rgb_s, dens = mlp(opt, params_f, points_from_depth(center, ray, t_all), ray, mode, noise_f, fine=True)
        fine = dict(rgb_samples=rgb_s, density_samples=dens, t=t_all)
        fine.update(composite(opt, ray, rgb_s, dens, t_all))
        out.update({k + "_fine": v for k, v in fine.items()})
    return out


def render_to_max(opt, params_c, params_f, center, ray, depth_min, depth_max, mode=None, it=None,
                  noise_c=None, noise_f=None):
    """renderer.py:504-593: deterministic samples up to a per-ray far bound, the
    fine network evaluated on the SAME samples; extra iteration gate :579-581."""
    Nc = opt.nerf.sample_intvs
    t = sample_depth_to_max(Nc, depth_min, depth_max, ray.dtype)
    out = dict(origins=center, viewdirs=ray)
    rgb_s, dens = mlp(opt, params_c, points_from_depth(center, ray, t), ray, mode, noise_c)
    out.update(rgb_samples=rgb_s, density_samples=dens, t=t)
    out.update(composite(opt, ray, rgb_s, dens, t))
    skip = _fine_gate_skips(opt, it)
    s = getattr(opt.nerf, "start_fine_sampling_at_x", None) if not hasattr(opt.nerf, "get") else opt.nerf.get("start_fine_sampling_at_x", None)
    if not skip and s is not None and it is not None and it < s:
        skip = True
    if opt.nerf.fine_sampling and not skip:
        rgb_s, dens = mlp(opt, params_f, points_from_depth(center, ray, t), ray, mode, noise_f, fine=True)
        fine = dict(rgb_samples=rgb_s, density_samples=dens, t=t)
        fine.update(composite(opt, ray, rgb_s, dens, t))
        out.update({k + "_fine": v for k, v in fine.items()})
    return out
