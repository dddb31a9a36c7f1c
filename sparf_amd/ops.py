"""torch <-> libsparf_hip glue: tensor allocation, stream hand-off and the autograd
boundary.  PyTorch is plumbing here (device memory, current stream, autograd graph edges);
every FLOP of the hot path runs in the HIP kernels behind the C ABI.
"""
import ctypes
import math
from ctypes import c_void_p

import torch

from . import lib as L


_C2F_OFF = {}


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def pack_weights(params, prec, out=None):
    """params: 20 tensors (W0,b0,...,W9,b9) in nn.Linear layout on one cuda device.
    Returns the packed uint8 blob consumed by the pass kernels."""
    prec = L.base_prec(prec)
    lib = L.load()
    dev = params[0].device
    L.require_gpu(dev)
    for p, (o, i) in zip(params[0::2], L.LAYER_SHAPES):
        if tuple(p.shape) != (o, i):
            raise L.SparfError(f"unsupported architecture: weight {tuple(p.shape)} where {(o, i)} is compiled in "
                               "(8x256 feature MLP with skip at 4, 128-wide colour branch, L_3D=10, L_view=4)")
    ps = [_f32(p) for p in params]
    arr = (c_void_p * 20)(*[p.data_ptr() for p in ps])
    nbytes = lib.sparf_packed_bytes(prec)
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    tables = L.tables_device(L.base_prec(prec), dev)
    with L.on(dev):
        L.check(lib.sparf_pack_weights(prec, arr, L.ptr(tables), L.ptr(out), L.stream_ptr(dev)), "sparf_pack_weights")
    return out


def c2f_weights(progress, barf_c2f, device):
    """The 16-float band-weight vector of one pass (frequency_nerf.py:248-253), computed on the
    device from the CURRENT value of `progress` -- never cached: the reference trainer rewrites
    progress.data every iteration (nerf_trainer.py:273-275), which no version counter sees."""
    lib = L.load()
    L.require_gpu(device)
    has = barf_c2f is not None
    if not has:                       # no masking: the constant vector, made once per device
        dev = torch.device(device)
        if dev.index is None:         # a bare "cuda" must not bind the cache entry to whichever device was current first
            dev = torch.device("cuda", torch.cuda.current_device())
        if dev not in _C2F_OFF:       # built on the device (no pageable host->device copy: legal under stream capture)
            v = torch.ones(16, dtype=torch.float32, device=dev)
            v[14:] = 0.0
            _C2F_OFF[dev] = v
        return _C2F_OFF[dev]
    out = torch.empty(16, dtype=torch.float32, device=device)
    prog = _f32(progress).reshape(1).to(device)
    s, e = (float(barf_c2f[0]), float(barf_c2f[1])) if has else (0.0, 1.0)
    with L.on(device):
        L.check(lib.sparf_c2f_weights(L.ptr(prog), int(has), s, e, L.ptr(out), L.stream_ptr(device)), "sparf_c2f_weights")
    return out


def _resolve(device):
    """torch.device with an index: the reference trainer hands Graph a bare 'cuda' (base_trainer.py:109), tensors made on it live
    on 'cuda:<current>'"""
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def _check_out(out, shape, device):
    if out.dtype != torch.float32 or tuple(out.shape) != tuple(shape) or not out.is_contiguous() or _resolve(out.device) != _resolve(device):
        raise L.SparfError(f"out= must be a dense float32 {tuple(shape)} tensor on {device}")
    return out


def sample_coarse(nrays, nsamp, dmin, scale, inverse, device, jitter=None, u_const=0.5, dmax_ray=None, range_dev=None, out=None):
    """range_dev: optional float32 device tensor {dmin, dmax} (or {dmin}) replacing the floats.
    out: optional [nrays, nsamp] destination (a row slice of a shared ray buffer, Graph.render_batch)."""
    lib = L.load()
    L.require_gpu(device)
    t = torch.empty(nrays, nsamp, dtype=torch.float32, device=device) if out is None else _check_out(out, (nrays, nsamp), device)
    j = _f32(jitter).reshape(nrays, nsamp) if jitter is not None else None
    dm = _f32(dmax_ray).reshape(nrays) if dmax_ray is not None else None
    with L.on(device):
        L.check(lib.sparf_sample_coarse(L.ptr(j), float(u_const), L.ptr(dm), L.ptr(range_dev), float(dmin), float(scale),
                                        int(bool(inverse)), nrays, nsamp, L.ptr(t), L.stream_ptr(device)), "sparf_sample_coarse")
    return t


def sample_fine(weights, t_coarse, u_mid, dmin, dmax, want_unsorted=False, range_dev=None, out=None):
    """weights, t_coarse [R, Nc]; u_mid [Nf].  Returns (sorted union [R, Nc+Nf], t_fine or None).
    out: optional [R, Nc+Nf] destination for the sorted union."""
    lib = L.load()
    dev = weights.device
    L.require_gpu(dev)
    R, Nc = weights.shape
    Nf = u_mid.numel()
    w, tc, um = _f32(weights), _f32(t_coarse), _f32(u_mid)
    out = torch.empty(R, Nc + Nf, dtype=torch.float32, device=dev) if out is None else _check_out(out, (R, Nc + Nf), dev)
    tf = torch.empty(R, Nf, dtype=torch.float32, device=dev) if want_unsorted else None
    # a HOST grid (the reference draws it on the CPU, renderer.py:439) travels in the launch arguments: no host -> device copy
    fn = lib.sparf_sample_fine if um.device.type == "cuda" else lib.sparf_sample_fine_hostgrid
    if um.device.type != "cuda" and Nf > 256:
        um, fn = um.to(dev), lib.sparf_sample_fine
    with L.on(dev):
        L.check(fn(L.ptr(w), L.ptr(tc), L.ptr(um), L.ptr(range_dev), float(dmin), float(dmax), R, Nc, Nf, L.ptr(tf),
                   L.ptr(out), L.stream_ptr(dev)), "sparf_sample_fine")
    return out, tf


def _segments(segs, grads=None):
    """ctypes array of sparf_segment_t from [(ray0, nrays, noise_scale), ...] (+ per-segment upstream gradients)"""
    if len(segs) > L.MAX_SEGMENTS:
        raise L.SparfError(f"at most {L.MAX_SEGMENTS} ray segments per pass")
    arr = (L.Segment * len(segs))()
    for i, (r0, n, ns) in enumerate(segs):
        arr[i].ray0, arr[i].nrays, arr[i].noise_scale = int(r0), int(n), float(ns)
        if grads is not None:
            g = [x.data_ptr() if x is not None else None for x in grads[i]]
            (arr[i].g_rgb, arr[i].g_depth, arr[i].g_opacity, arr[i].g_weights, arr[i].g_depth_var, arr[i].g_rgb_var, arr[i].g_all_cumulated,
             arr[i].g_density, arr[i].g_rgb_samples) = g + [None] * (9 - len(g))
    return arr


def build_pass_fwd(prec, c, d, tt, nz, noise_scale, white_bg, packed, c2f, save, segs=None, far=None):
    """Allocate outputs and fill the C struct of sparf_pass_forward.  Returns
    (struct, outputs dict, save buffer(s) or None, scratch list to keep alive).
    segs: optional [(ray0, nrays, noise_scale), ...] ray segments (include/sparf_hip.h sparf_segment_t).
    far: optional (K, far_prec, far_packed): the last K samples of every ray also run through far_prec (sparf_hip.h "far rows");
    what that launch saves is transplanted into `save` by the call, its own save area is scratch (kept alive in the returned list
    until the stream has run the call: the caching allocator hands it out again in stream order)."""
    lib = L.load()
    dev = c.device
    R, N = tt.shape
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    out = dict(raylen=f(R), sigma_raw=f(R, N), rgb_samples=f(R, N, 3), density=f(R, N), weights=f(R, N), rgb=f(R, 3),
               depth=f(R), opacity=f(R), depth_var=f(R), rgb_var=f(R), all_cumulated=f(R))
    save_buf = torch.empty(lib.sparf_save_bytes(prec, R * N), dtype=torch.uint8, device=dev) if save else None
    venc = torch.empty(R * 32 * (2 if L.base_prec(prec) == L.PREC_BF16 else 4), dtype=torch.uint8, device=dev)
    a = L.PassFwd(prec=prec, nrays=R, nsamp=N, center=c.data_ptr(), dir=d.data_ptr(), t=tt.data_ptr(),
                  noise=nz.data_ptr() if nz is not None else None, noise_scale=float(noise_scale), white_bg=int(bool(white_bg)),
                  packed=packed.data_ptr(), c2f=c2f.data_ptr(), save=save_buf.data_ptr() if save_buf is not None else None, venc_ws=venc.data_ptr(),
                  **{k: v.data_ptr() for k, v in out.items()})
    keep = [venc]
    if segs:
        sa = _segments(segs)
        a.nseg, a.seg = len(segs), sa
        keep.append(sa)
    if far is not None and isinstance(far[0], float):
        # far TILES by value: (threshold, far_prec, far_packed); inference passes only (sparf_hip.h far_count = -1)
        thr, fprec, fpacked = far
        if save or N % 32 != 0:
            raise L.SparfError("far tiles by value: inference passes with a multiple of 32 samples per ray only")
        fvenc = torch.empty(R * 32 * (2 if fprec == L.PREC_BF16 else 4), dtype=torch.uint8, device=dev)
        a.far_count, a.far_prec, a.far_packed, a.far_thr, a.far_venc_ws = -1, int(fprec), fpacked.data_ptr(), float(thr), fvenc.data_ptr()
        keep += [fpacked, fvenc]
    elif far is not None:
        K, fprec, fpacked = far
        if not 0 < K < N:
            raise L.SparfError(f"far rows: 0 < K < samples per ray, got K = {K} of {N}")
        far_ws = torch.empty(lib.sparf_save_bytes(fprec, R * K), dtype=torch.uint8, device=dev) if save else None
        a.far_count, a.far_prec, a.far_packed = int(K), int(fprec), fpacked.data_ptr()
        a.far_ws = far_ws.data_ptr() if far_ws is not None else None
        if fprec != prec:          # the view-encoding rows are laid out per precision: the far launch gets its own
            fvenc = torch.empty(R * 32 * (2 if fprec == L.PREC_BF16 else 4), dtype=torch.uint8, device=dev)
            a.far_venc_ws = fvenc.data_ptr()
            keep.append(fvenc)
        keep += [fpacked, far_ws]
    return a, out, save_buf, keep


def build_pass_bwd(prec, c, d, tt, nz, noise_scale, white_bg, packed, c2f, save, fwd_out, grads, pose, segs=None):
    """Allocate workspace / results and fill the C struct of sparf_pass_backward.
    grads = (g_rgb, g_depth, g_opacity, g_weights[, g_depth_var, g_rgb_var, g_all_cumulated, g_density, g_rgb_samples]), any may be
    None; with `segs` a list of such tuples, one per ray segment (each tensor covering only its segment's rays)."""
    lib = L.load()
    dev = c.device
    R, N = tt.shape
    ws = torch.empty(lib.sparf_bwd_workspace_bytes(prec, R, N, int(pose)), dtype=torch.uint8, device=dev)
    gp = torch.empty(L.N_PARAMS, dtype=torch.float32, device=dev)
    dc = torch.empty(R, 3, dtype=torch.float32, device=dev) if pose else None
    dd = torch.empty(R, 3, dtype=torch.float32, device=dev) if pose else None
    if segs:
        gseg = [[_f32(g) if g is not None else None for g in gt] for gt in grads]
        gs = [None] * 9
    else:
        gs = [_f32(g) if g is not None else None for g in grads]
        gs += [None] * (9 - len(gs))
    tables = L.tables_device(L.base_prec(prec), dev)
    P = lambda x: x.data_ptr() if x is not None else None
    a = L.PassBwd(prec=prec, nrays=R, nsamp=N, center=P(c), dir=P(d), t=P(tt), noise=P(nz), noise_scale=float(noise_scale),
                  white_bg=int(bool(white_bg)), packed=P(packed), c2f=P(c2f), tables=P(tables), save=P(save), raylen=P(fwd_out["raylen"]),
                  sigma_raw=P(fwd_out["sigma_raw"]), rgb_samples=P(fwd_out["rgb_samples"]), weights=P(fwd_out["weights"]),
                  g_rgb=P(gs[0]), g_depth=P(gs[1]), g_opacity=P(gs[2]), g_weights=P(gs[3]), ws=P(ws), grad_params=P(gp),
                  d_center=P(dc), d_dir=P(dd), g_depth_var=P(gs[4]), g_rgb_var=P(gs[5]), g_all_cumulated=P(gs[6]), g_density=P(gs[7]),
                  g_rgb_samples=P(gs[8]))
    keep = [ws, tables] + gs
    if segs:
        sa = _segments(segs, gseg)
        a.nseg, a.seg = len(segs), sa
        keep += [sa, gseg]
    return a, gp, dc, dd, keep


class NerfPass(torch.autograd.Function):
    """One network (coarse or fine) over R rays x N samples: fused MLP + compositing.

    Inputs  center [R,3], dirs [R,3], t [R,N], noise [R,N] | None, then the 20 parameter
            tensors (only used to route gradients; the kernels read `packed`).
    Outputs rgb [R,3], depth [R], opacity [R], weights [R,N], depth_var [R], rgb_var [R], all_cumulated [R], density [R,N],
            rgb_samples [R,N,3] -- ALL differentiable, as in the reference, where NeRF.composite is plain autograd
            (frequency_nerf.py:317-338; C ABI 6.  Rounds 1-4 marked the last five non-differentiable: a loss on depth_var
            silently received a zero gradient, VERDICT r04 missing-3).
    """

    @staticmethod
    def forward(ctx, center, dirs, t, noise, noise_scale, white_bg, prec, packed, c2f, grad_mode, far, *params):
        lib = L.load()
        dev = center.device
        L.require_gpu(dev)
        c, d, tt = _f32(center), _f32(dirs), _f32(t)
        nz = _f32(noise) if noise is not None else None
        # needs_input_grad ignores the caller's grad mode (and forward() itself always runs with
        # grad disabled): `grad_mode` = torch.is_grad_enabled() at the call site.  Without it
        # nothing is saved and the inference kernel runs.
        need_grad = bool(grad_mode) and any(ctx.needs_input_grad)
        ctx.set_materialize_grads(False)          # absent upstream gradients arrive as None, not as zero tensors
        a, out, save, _keep = build_pass_fwd(prec, c, d, tt, nz, noise_scale, white_bg, packed, c2f, need_grad, far=far)
        with L.on(dev):
            L.check(lib.sparf_pass_forward(ctypes.byref(a), L.stream_ptr(dev)), "sparf_pass_forward")
        if need_grad:       # (far rows leave nothing of their own behind: their saves were transplanted into `save`)
            ctx.save_for_backward(c, d, tt, nz, packed, c2f, save, out["raylen"], out["sigma_raw"], out["rgb_samples"], out["weights"])
            ctx.meta = (float(noise_scale), int(bool(white_bg)), prec, [tuple(p.shape) for p in params])
        return (out["rgb"], out["depth"], out["opacity"], out["weights"], out["depth_var"], out["rgb_var"], out["all_cumulated"],
                out["density"], out["rgb_samples"])

    @staticmethod
    def backward(ctx, *g9):
        lib = L.load()
        c, d, tt, nz, packed, c2f, save, raylen, sigma_raw, rgb_samples, weights = ctx.saved_tensors
        noise_scale, white_bg, prec, shapes = ctx.meta
        pose = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        fwd_out = dict(raylen=raylen, sigma_raw=sigma_raw, rgb_samples=rgb_samples, weights=weights)
        a, gp, dc, dd, _keep = build_pass_bwd(prec, c, d, tt, nz, noise_scale, white_bg, packed, c2f, save, fwd_out, g9, pose)
        with L.on(c.device):
            L.check(lib.sparf_pass_backward(ctypes.byref(a), L.stream_ptr(c.device)), "sparf_pass_backward")
        grads, off = [], 0
        for i, shp in enumerate(shapes):
            n = 1
            for s in shp:
                n *= s
            grads.append(gp[off:off + n].view(shp) if ctx.needs_input_grad[11 + i] else None)
            off += n
        return (dc if ctx.needs_input_grad[0] else None, dd if ctx.needs_input_grad[1] else None, None, None, None, None, None,
                None, None, None, None, *grads)


class NerfPassSeg(torch.autograd.Function):
    """NerfPass over the rays of SEVERAL render calls laid back to back: one launch set,
    the nine outputs returned PER SEGMENT as views of the pass's buffers (no split copies), the upstream
    gradients consumed per segment through the C ABI's segment table (no gather copies).
    segs = [(ray0, nrays, noise_scale), ...]; noise: one [R,N] tensor or None."""

    @staticmethod
    def forward(ctx, center, dirs, t, noise, white_bg, prec, packed, c2f, grad_mode, segs, far, *params):
        lib = L.load()
        dev = center.device
        L.require_gpu(dev)
        c, d, tt = _f32(center), _f32(dirs), _f32(t)
        nz = _f32(noise) if noise is not None else None
        need_grad = bool(grad_mode) and any(ctx.needs_input_grad)
        ctx.set_materialize_grads(False)
        a, out, save, _keep = build_pass_fwd(prec, c, d, tt, nz, 0.0, white_bg, packed, c2f, need_grad, segs=segs, far=far)
        with L.on(dev):
            L.check(lib.sparf_pass_forward(ctypes.byref(a), L.stream_ptr(dev)), "sparf_pass_forward")
        if need_grad:
            ctx.save_for_backward(c, d, tt, nz, packed, c2f, save, out["raylen"], out["sigma_raw"], out["rgb_samples"], out["weights"])
            ctx.meta = (int(bool(white_bg)), prec, [tuple(p.shape) for p in params], list(segs))
        keys = ("rgb", "depth", "opacity", "weights", "depth_var", "rgb_var", "all_cumulated", "density", "rgb_samples")
        res = []
        for (r0, n, _) in segs:
            res += [out[k][r0:r0 + n] for k in keys]
        return tuple(res)

    @staticmethod
    def backward(ctx, *g):
        lib = L.load()
        c, d, tt, nz, packed, c2f, save, raylen, sigma_raw, rgb_samples, weights = ctx.saved_tensors
        white_bg, prec, shapes, segs = ctx.meta
        pose = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        fwd_out = dict(raylen=raylen, sigma_raw=sigma_raw, rgb_samples=rgb_samples, weights=weights)
        gseg = [tuple(g[9 * i:9 * i + 9]) for i in range(len(segs))]
        a, gp, dc, dd, _keep = build_pass_bwd(prec, c, d, tt, nz, 0.0, white_bg, packed, c2f, save, fwd_out, gseg, pose, segs=segs)
        with L.on(c.device):
            L.check(lib.sparf_pass_backward(ctypes.byref(a), L.stream_ptr(c.device)), "sparf_pass_backward")
        grads, off = [], 0
        for i, shp in enumerate(shapes):
            n = 1
            for s in shp:
                n *= s
            grads.append(gp[off:off + n].view(shp) if ctx.needs_input_grad[11 + i] else None)
            off += n
        return (dc if ctx.needs_input_grad[0] else None, dd if ctx.needs_input_grad[1] else None, None, None, None, None, None,
                None, None, None, None, *grads)


def nerf_pass_segments(center, dirs, t, noise, white_bg, prec, packed, c2f, params, segs, far=None):
    """-> list (one per segment) of dicts with the reference's composite keys (flat ray axis)"""
    flat = NerfPassSeg.apply(center, dirs, t, noise, white_bg, prec, packed, c2f, torch.is_grad_enabled(), list(segs), far, *params)
    keys = ("rgb", "depth", "opacity", "weights", "depth_var", "rgb_var", "all_cumulated", "density_samples", "rgb_samples")
    return [dict(zip(keys, flat[9 * i:9 * i + 9])) for i in range(len(segs))]


def nerf_pass(center, dirs, t, noise, noise_scale, white_bg, prec, packed, c2f, params, far=None):
    """Convenience wrapper returning a dict with the reference's composite keys (flat ray axis).
    c2f: the pass's band-weight vector (c2f_weights).  far: (K, far_prec, far_packed) or None (build_pass_fwd)."""
    rgb, depth, opacity, weights, depth_var, rgb_var, all_cum, density, rgb_s = NerfPass.apply(
        center, dirs, t, noise, noise_scale, white_bg, prec, packed, c2f, torch.is_grad_enabled(), far, *params)
    return dict(rgb=rgb, depth=depth, opacity=opacity, weights=weights, depth_var=depth_var, rgb_var=rgb_var,
                all_cumulated=all_cum, density_samples=density, rgb_samples=rgb_s)


def _ray_sel(pose, pixels, ray_idx):
    """(sel tensor, is_pixels, per_image, N) of a ray-generation request"""
    B = pose.shape[0]
    if pixels is not None:
        sel = _f32(pixels)
        per_image, N = int(sel.dim() == 3), sel.shape[-2]
    else:
        sel = ray_idx.to(device=pose.device, dtype=torch.int64).contiguous()
        per_image, N = int(sel.dim() == 2), sel.shape[-1]
    if per_image and sel.shape[0] != B:
        raise ValueError("per-image pixels / ray_idx must have one row per pose")
    return sel, pixels is not None, per_image, N


def _ray_gen_launch(P, K, sel, is_px, per_image, width, B, N, center, ray):
    lib = L.load()
    Pp = L.ptr
    with L.on(P.device):
        L.check(lib.sparf_ray_gen_forward(Pp(P), Pp(K), Pp(sel if is_px else None), Pp(None if is_px else sel), per_image, int(width), B, N,
                                          Pp(center), Pp(ray), L.stream_ptr(P.device)), "sparf_ray_gen_forward")


def _ray_gen_pose_grad(P, K, sel, is_px, per_image, width, B, N, g_center, g_ray):
    if g_center is None and g_ray is None:
        return None
    lib = L.load()
    gc = _f32(g_center) if g_center is not None else None
    gr = _f32(g_ray) if g_ray is not None else None
    d_pose = torch.empty(B, 3, 4, device=P.device, dtype=torch.float32)
    Pp = L.ptr
    with L.on(P.device):
        L.check(lib.sparf_ray_gen_backward(Pp(P), Pp(K), Pp(sel if is_px else None), Pp(None if is_px else sel), per_image, int(width), B, N,
                                           Pp(gc), Pp(gr), Pp(d_pose), L.stream_ptr(P.device)), "sparf_ray_gen_backward")
    return d_pose


def _ray_gen_pixel_grad(P, K, per_image, B, N, g_ray, pix_shape):
    """d loss / d pixels of ray generation: ray = R^T K^-1 [x, y, 1] (camera.py:296-306, 321-326, 400-406), so
    d/d(x, y) = (K^-1[:, :2])^T R d/d ray; the ray origin does not depend on the pixel.  The reference's ray generation is
    plain autograd and the depth-consistency loss hands it pixel coordinates that DO carry a gradient (projections of points
    back-projected with a rendered depth, depth_cons_loss.py:199-201, 254-262 -> :291: nothing is detached), so the gradient
    flows on into the reference render that produced that depth.  Two small batched matrix products in PyTorch: K^-1 in
    float64 as the fused kernel uses it."""
    if g_ray is None:
        return None
    from .camera import _intr_inverse
    R = P[:, :, :3]
    Kinv = _intr_inverse(K)
    dg = g_ray.reshape(B, N, 3) @ R.transpose(-1, -2)              # R . d ray, as row vectors
    dxy = dg @ Kinv[:, :, :2]                                      # [B, N, 2]
    return dxy.reshape(pix_shape) if per_image else dxy.sum(0).reshape(pix_shape)


class RayGen(torch.autograd.Function):
    """Ray origins / directions for the selected pixels of every image in one launch
    (replaces camera.get_center_and_ray[_at_pixels], camera.py:347-416).

    pose [B,3,4] w2c (differentiable), intr [B,3,3] (no gradient), pixels [N,2] | [B,N,2]
    float (x, y) OR ray_idx [N] | [B,N] int64 flat indices (pixel centres, +0.5).
    Returns center, ray [B,N,3]."""

    @staticmethod
    def forward(ctx, pose, intr, pixels, ray_idx, width):
        dev = pose.device
        L.require_gpu(dev)
        B = pose.shape[0]
        P, K = _f32(pose), _f32(intr)
        sel, is_px, per_image, N = _ray_sel(pose, pixels, ray_idx)
        center = torch.empty(B, N, 3, device=dev, dtype=torch.float32)
        ray = torch.empty(B, N, 3, device=dev, dtype=torch.float32)
        _ray_gen_launch(P, K, sel, is_px, per_image, width, B, N, center, ray)
        ctx.save_for_backward(P, K, sel)
        ctx.meta = (is_px, per_image, int(width), B, N)
        ctx.pix_shape = tuple(pixels.shape) if pixels is not None else None
        ctx.set_materialize_grads(False)
        return center, ray

    @staticmethod
    def backward(ctx, g_center, g_ray):
        P, K, sel = ctx.saved_tensors
        is_px, per_image, width, B, N = ctx.meta
        d_pose = _ray_gen_pose_grad(P, K, sel, *ctx.meta, g_center, g_ray) if ctx.needs_input_grad[0] else None
        d_pix = _ray_gen_pixel_grad(P, K, per_image, B, N, g_ray, ctx.pix_shape) if (is_px and ctx.needs_input_grad[2]) else None
        return d_pose, None, d_pix, None, None


def ray_gen(pose, intr, pixels=None, ray_idx=None, width=0):
    return RayGen.apply(pose, intr, pixels, ray_idx, width)


class RayGenMany(torch.autograd.Function):
    """Ray generation of SEVERAL render requests into ONE [2, R_total, 3] (centres, directions) buffer, each request
    at its row offset (Graph.render_batch): one autograd node, no concatenation; the backward hands
    every request's pose its own gradient from that request's rows of the buffer gradient.
    specs: [(intr, pixels, ray_idx, width), ...]; poses: the matching [B_i,3,4] tensors.
    (Writing the requests into views of a shared buffer with in-place autograd was tried first: a custom Function that
    marks a VIEW dirty drops the history an earlier in-place write gave the base -- the first request's pose lost its
    gradient; reproduced on CPU with plain torch ops around it.)"""

    @staticmethod
    def forward(ctx, specs, *poses):
        dev = poses[0].device
        L.require_gpu(dev)
        reqs, total = [], 0
        for (intr, pixels, ray_idx, width), pose in zip(specs, poses):
            sel, is_px, per_image, N = _ray_sel(pose, pixels, ray_idx)
            B = pose.shape[0]
            reqs.append((_f32(pose), _f32(intr), sel, (is_px, per_image, int(width), B, N), total))
            total += B * N
        rays = torch.empty(2, total, 3, device=dev, dtype=torch.float32)
        for P, K, sel, meta, off in reqs:
            n = meta[3] * meta[4]
            if n > 0:
                _ray_gen_launch(P, K, sel, *meta, rays[0, off:off + n], rays[1, off:off + n])
        ctx.save_for_backward(*[t for r in reqs for t in r[:3]])
        ctx.metas = [(r[3], r[4]) for r in reqs]
        ctx.set_materialize_grads(False)
        return rays

    @staticmethod
    def backward(ctx, g):
        out = [None]
        saved = ctx.saved_tensors
        for i, (meta, off) in enumerate(ctx.metas):
            n = meta[3] * meta[4]
            if g is None or n == 0 or not ctx.needs_input_grad[1 + i]:
                out.append(None)
                continue
            P, K, sel = saved[3 * i:3 * i + 3]
            out.append(_ray_gen_pose_grad(P, K, sel, *meta, g[0, off:off + n], g[1, off:off + n]))
        return tuple(out)


def ray_gen_many(specs, poses):
    """-> rays [2, R_total, 3]; request i occupies rows [off_i, off_i + B_i * N_i) in request order"""
    return RayGenMany.apply(list(specs), *poses)


class PhotometricLoss(torch.autograd.Function):
    """MSE_loss / huber_loss (delta 0.5, x2) of base_losses.py:151-156 on rgb and optionally
    rgb_fine against one target, summed (base_losses.py:303-311): one launch for the loss
    and its gradient seed."""

    @staticmethod
    def forward(ctx, rgb, rgb_fine, target, kind, delta):
        lib = L.load()
        dev = rgb.device
        L.require_gpu(dev)
        p, t = _f32(rgb), _f32(target)
        pf = _f32(rgb_fine) if rgb_fine is not None else None
        if t.numel() != p.numel() or (pf is not None and pf.numel() != p.numel()):
            raise ValueError("photometric_loss: rgb, rgb_fine and target must have the same number of elements")
        loss = torch.empty((), device=dev, dtype=torch.float32)
        need, need_f = ctx.needs_input_grad[0], pf is not None and ctx.needs_input_grad[1]
        d = torch.empty_like(p) if need else None
        df = torch.empty_like(pf) if need_f else None
        P = L.ptr
        ws = torch.empty(int(lib.sparf_photometric_workspace_floats()), device=dev, dtype=torch.float32) if p.numel() > 65536 else None
        with L.on(dev):
            L.check(lib.sparf_photometric_loss(P(p), P(pf), P(t), p.numel(), int(kind), float(delta), P(loss), P(d), P(df), P(ws),
                                               L.stream_ptr(dev)), "sparf_photometric_loss")
        ctx.save_for_backward(d, df)
        ctx.shapes = (rgb.shape, rgb_fine.shape if rgb_fine is not None else None)
        return loss

    @staticmethod
    def backward(ctx, g):
        d, df = ctx.saved_tensors
        return (d.view(ctx.shapes[0]) * g if d is not None else None, df.view(ctx.shapes[1]) * g if df is not None else None,
                None, None, None)


def photometric_loss(rgb, target, rgb_fine=None, huber=False, delta=0.5):
    return PhotometricLoss.apply(rgb, rgb_fine, target, 1 if huber else 0, delta)


# ---------------------------------------------------------------------------------------------- stand-alone compositing
class Composite(torch.autograd.Function):
    """NeRF.composite (frequency_nerf.py:283-343) on caller-built per-sample values (C ABI 6 sparf_composite_forward / _backward):
    ray [R,3], density [R,N] (after softplus), rgb_samples [R,N,3] (after the sigmoid), t [R,N].
    -> rgb [R,3], depth, opacity [R], weights [R,N], depth_var, rgb_var, all_cumulated [R]: all differentiable w.r.t. density,
    rgb_samples and the ray (through its length: dist = delta * |ray|, :302-308); the depth samples receive none."""

    @staticmethod
    def forward(ctx, ray, density, rgb_samples, t, white_bg):
        lib = L.load()
        dev = ray.device
        L.require_gpu(dev)
        d, dn, cs, tt = _f32(ray), _f32(density), _f32(rgb_samples), _f32(t)
        R, N = tt.shape
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        out = dict(raylen=f(R), weights=f(R, N), rgb=f(R, 3), depth=f(R), opacity=f(R), depth_var=f(R), rgb_var=f(R), all_cumulated=f(R))
        a = L.CompositeFwd(nrays=R, nsamp=N, white_bg=int(bool(white_bg)), dir=d.data_ptr(), t=tt.data_ptr(), density=dn.data_ptr(),
                           rgb_samples=cs.data_ptr(), **{k: v.data_ptr() for k, v in out.items()})
        with L.on(dev):
            L.check(lib.sparf_composite_forward(ctypes.byref(a), L.stream_ptr(dev)), "sparf_composite_forward")
        ctx.save_for_backward(d, dn, cs, tt, out["raylen"], out["weights"])
        ctx.white_bg = int(bool(white_bg))
        ctx.set_materialize_grads(False)
        return out["rgb"], out["depth"], out["opacity"], out["weights"], out["depth_var"], out["rgb_var"], out["all_cumulated"]

    @staticmethod
    def backward(ctx, *g7):
        lib = L.load()
        d, dn, cs, tt, raylen, weights = ctx.saved_tensors
        dev = d.device
        R, N = tt.shape
        gs = [_f32(g) if g is not None else None for g in g7]
        d_dens = torch.empty(R, N, dtype=torch.float32, device=dev)
        d_rgbs = torch.empty(R, N, 3, dtype=torch.float32, device=dev)
        need_ray = ctx.needs_input_grad[0]
        d_dir = torch.empty(R, 3, dtype=torch.float32, device=dev) if need_ray else None
        d_len = torch.empty(R, dtype=torch.float32, device=dev) if need_ray else None
        P = lambda x: x.data_ptr() if x is not None else None
        a = L.CompositeBwd(nrays=R, nsamp=N, white_bg=ctx.white_bg, dir=P(d), t=P(tt), density=P(dn), rgb_samples=P(cs), raylen=P(raylen),
                           weights=P(weights), g_rgb=P(gs[0]), g_depth=P(gs[1]), g_opacity=P(gs[2]), g_weights=P(gs[3]), g_depth_var=P(gs[4]),
                           g_rgb_var=P(gs[5]), g_all_cumulated=P(gs[6]), d_density=P(d_dens), d_rgb_samples=P(d_rgbs), d_dir=P(d_dir), d_len_ws=P(d_len))
        with L.on(dev):
            L.check(lib.sparf_composite_backward(ctypes.byref(a), L.stream_ptr(dev)), "sparf_composite_backward")
        return d_dir, d_dens if ctx.needs_input_grad[1] else None, d_rgbs if ctx.needs_input_grad[2] else None, None, None


def composite(ray, density, rgb_samples, t, white_bg):
    """-> dict with the reference's composite keys (flat ray axis)"""
    if t.requires_grad and torch.is_grad_enabled():
        raise L.SparfError("composite: depth samples that require a gradient are not supported (none of the reference's callers "
                           "differentiates them: renderer.py:323 no_grad, :405-407 fresh draws)")
    rgb, depth, opacity, weights, depth_var, rgb_var, all_cum = Composite.apply(ray, density, rgb_samples, t, white_bg)
    return dict(rgb=rgb, depth=depth, opacity=opacity, weights=weights, depth_var=depth_var, rgb_var=rgb_var, all_cumulated=all_cum)


# ---------------------------------------------------------------------------------------------- one render call = one autograd node
# Graph.render (renderer.py:250-345) behind ONE autograd.Function: coarse depths -> coarse pass -> resampling + merge -> fine pass, issued
# from one Python frame through the C ABI with every fp32 result of BOTH passes in ONE allocation (an "arena": the outputs are views of
# it), the per-pass C structs filled from cached offset tables, and one backward that runs both passes' backward calls and hands
# the parameter gradients over as views of one [2, N_PARAMS] buffer.  Round 4 ran a render as two NerfPass nodes + ~15 allocations
# + ~40 ctypes attribute stores per pass: 1.2 ms of host time per render + backward (profiles/r04_host_overhead.log), which is what
# the six render calls per iteration of the unmodified SPARF losses were bounded by.
_PASS_F32 = (("raylen", 1, 0), ("sigma_raw", 0, 1), ("rgb_samples", 0, 3), ("density", 0, 1), ("weights", 0, 1), ("rgb", 3, 0), ("depth", 1, 0),
             ("opacity", 1, 0), ("depth_var", 1, 0), ("rgb_var", 1, 0), ("all_cumulated", 1, 0), ("t", 0, 1))     # (name, floats per ray, floats per sample)
_PARAM_SIZES = [n for (o, i) in L.LAYER_SHAPES for n in (o * i, o)]
_PLANS = {}


class _RenderPlan:
    """offsets (in floats) of every fp32 result of a render inside its two arenas, for R rays, Nc coarse and Nf fine samples.  Arena 0
    holds what is per RAY (rgb, depth, opacity, the variances, all_cumulated, |ray|, the band weights), arena 1 what is per SAMPLE
    (depths, raw density, colours, density, weights): a caller that keeps `ret.rgb` of a slice alive -- Graph.render_by_slices does, until
    its final cat; so does logging code -- pins a few floats per ray, not the ~28 bytes per sample row and pass of the whole render
    (a 756 x 1008 evaluation image retained 5.5 GB that way when both lived in one allocation)."""

    def __init__(self, R, Nc, Nf):
        self.R, self.Nc, self.Nf = R, Nc, Nf
        self.off, self.order, tot = {}, ([], []), [0, 0]
        for tag, N in (("c", Nc), ("f", Nc + Nf)):
            if tag == "f" and Nf == 0:           # no fine pass (gated off / no fine network): nothing reserved for it
                continue
            for name, per_ray, per_samp in _PASS_F32:
                k = 0 if per_samp == 0 else 1
                n = R * (per_ray + per_samp * N)
                n_al = (n + 63) // 64 * 64                       # 256-byte granules
                self.off[tag + name] = (tot[k], n, k)
                self.order[k].append((tag + name, n_al))
                tot[k] += n_al
            self.off[tag + "c2f"] = (tot[0], 16, 0)
            self.order[0].append((tag + "c2f", 64))
            tot[0] += 64
        self.total = tuple(tot)


def _plan(R, Nc, Nf):
    key = (R, Nc, Nf)
    p = _PLANS.get(key)
    if p is None:
        if len(_PLANS) > 256:
            _PLANS.clear()
        p = _PLANS[key] = _RenderPlan(R, Nc, Nf)
    return p


def _contig32(g):
    return g if (g.dtype is torch.float32 and g.is_contiguous()) else g.detach().to(torch.float32).contiguous()


class RenderFn(torch.autograd.Function):
    """inputs : center, dirs [R,3] (differentiable); cfg (dict, below); jitter [R,Nc] | None, u_mid [Nf] | None, noise_c [R,Nc] | None,
                noise_f [R,Nc+Nf] | None, range_dev | None, packed_c, packed_f | None, far_packed_c, far_packed_f | None,
                prog_c, prog_f (the networks' `progress` scalars, device fp32: the BARF band weights of each pass are computed from
                their CURRENT device value inside this call, frequency_nerf.py:248-253; ignored when cfg['c2f'] is None); theta_c, theta_f:
                the networks' parameters as ONE flat autograd tensor each (NeRF.flat_params: the route of the parameter gradient -- the
                kernels read the packed streams) or None (no gradient wanted).
                u_mid may be a HOST tensor (the grid of renderer.py:439 is drawn on the CPU): it then travels in the launch arguments.
       cfg    : R, Nc, Nf, fine, dmin, dmax, scale, inverse, u_const, noise_scale, white_bg, prec_c, prec_f (pass precision ids), far_c, far_f
                ((K, far_prec) | None), c2f ((start, end) | None), grad (torch.is_grad_enabled() at the call site)
       outputs: per pass (coarse, then fine if cfg['fine']) rgb [R,3], depth, opacity [R], weights [R,N], depth_var, rgb_var, all_cumulated
                [R], density [R,N], rgb_samples [R,N,3] (differentiable) and t [R,N] (the pass's depth samples; no gradient)."""

    @staticmethod
    def forward(ctx, center, dirs, cfg, jitter, u_mid, noise_c, noise_f, range_dev, packed_c, packed_f, far_packed_c, far_packed_f, prog_c, prog_f, theta_c, theta_f):
        lib = L.load()
        dev = center.device
        R, Nc, Nf, fine = cfg["R"], cfg["Nc"], cfg["Nf"], cfg["fine"]
        plan = _plan(R, Nc, Nf if fine else 0)
        c, d = _contig32(center), _contig32(dirs)
        need_grad = bool(cfg["grad"]) and any(ctx.needs_input_grad)
        ctx.set_materialize_grads(False)
        arenas = (torch.empty(plan.total[0], dtype=torch.float32, device=dev), torch.empty(plan.total[1], dtype=torch.float32, device=dev))
        bases = (arenas[0].data_ptr(), arenas[1].data_ptr())
        A = lambda name: bases[plan.off[name][2]] + 4 * plan.off[name][0]
        stream = L.stream_ptr(dev)
        passes = [("c", Nc, cfg["prec_c"], cfg["far_c"], packed_c, far_packed_c, noise_c, prog_c)]
        if fine:
            passes.append(("f", Nc + Nf, cfg["prec_f"], cfg["far_f"], packed_f, far_packed_f, noise_f, prog_f))
        saves, keep = [], []
        c2f_off = None if cfg["c2f"] is not None else c2f_weights(None, None, dev).data_ptr()     # no masking: the constant vector of the device
        with L.on(dev):
            L.check(lib.sparf_sample_coarse(L.ptr(jitter), float(cfg["u_const"]), None, L.ptr(range_dev), float(cfg["dmin"]), float(cfg["scale"]),
                                            int(cfg["inverse"]), R, Nc, c_void_p(A("ct")), stream), "sparf_sample_coarse")
            for tag, N, prec, far, packed, far_packed, noise, prog in passes:
                if c2f_off is None:       # the pass's band weights, from the device value of its network's progress right now
                    if prog.dtype is not torch.float32 or prog.device != dev:
                        raise L.SparfError("NeRF.progress must be a float32 scalar on the renderer's device")
                    L.check(lib.sparf_c2f_weights(c_void_p(prog.data_ptr()), 1, float(cfg["c2f"][0]), float(cfg["c2f"][1]), c_void_p(A(tag + "c2f")), stream),
                            "sparf_c2f_weights")
                c2f_ptr = c2f_off if c2f_off is not None else A(tag + "c2f")
                if tag == "f":
                    fn = lib.sparf_sample_fine if u_mid.device.type == "cuda" else lib.sparf_sample_fine_hostgrid
                    L.check(fn(c_void_p(A("cweights")), c_void_p(A("ct")), L.ptr(u_mid), L.ptr(range_dev), float(cfg["dmin"]), float(cfg["dmax"]),
                               R, Nc, Nf, None, c_void_p(A("ft")), stream), "sparf_sample_fine")
                bp = L.base_prec(prec)
                save = torch.empty(lib.sparf_save_bytes(prec, R * N), dtype=torch.uint8, device=dev) if need_grad else None
                venc = torch.empty(R * 32 * (2 if bp == L.PREC_BF16 else 4), dtype=torch.uint8, device=dev)
                keep.append(venc)
                a = L.PassFwd(prec=prec, nrays=R, nsamp=N, center=c.data_ptr(), dir=d.data_ptr(), t=A(tag + "t"),
                              noise=noise.data_ptr() if noise is not None else None, noise_scale=float(cfg["noise_scale"]) if noise is not None else 0.0,
                              white_bg=int(cfg["white_bg"]), packed=packed.data_ptr(), c2f=c2f_ptr, save=save.data_ptr() if save is not None else None,
                              venc_ws=venc.data_ptr(), raylen=A(tag + "raylen"), sigma_raw=A(tag + "sigma_raw"), rgb_samples=A(tag + "rgb_samples"),
                              density=A(tag + "density"), weights=A(tag + "weights"), rgb=A(tag + "rgb"), depth=A(tag + "depth"), opacity=A(tag + "opacity"),
                              depth_var=A(tag + "depth_var"), rgb_var=A(tag + "rgb_var"), all_cumulated=A(tag + "all_cumulated"))
                if far is not None:
                    K, fprec = far
                    far_ws = torch.empty(lib.sparf_save_bytes(fprec, R * K), dtype=torch.uint8, device=dev) if need_grad else None
                    fvenc = torch.empty(R * 32 * (2 if fprec == L.PREC_BF16 else 4), dtype=torch.uint8, device=dev)
                    a.far_count, a.far_prec, a.far_packed = int(K), int(fprec), far_packed.data_ptr()
                    a.far_ws = far_ws.data_ptr() if far_ws is not None else None
                    a.far_venc_ws = fvenc.data_ptr()
                    keep += [far_ws, fvenc]
                L.check(lib.sparf_pass_forward(ctypes.byref(a), stream), "sparf_pass_forward")
                saves.append(save)
        # the results: views of the two arenas, one split per arena + one view each
        pieces = {}
        for k in (0, 1):
            pieces.update(zip([n for n, _ in plan.order[k]], arenas[k].split_with_sizes([s for _, s in plan.order[k]])))
        outs = []
        for tag, N, *_ in passes:
            g = lambda name, *shape: pieces[tag + name][:plan.off[tag + name][1]].view(*shape) if plan.off[tag + name][1] != pieces[tag + name].numel() \
                else pieces[tag + name].view(*shape)
            outs += [g("rgb", R, 3), g("depth", R), g("opacity", R), g("weights", R, N), g("depth_var", R), g("rgb_var", R), g("all_cumulated", R),
                     g("density", R, N), g("rgb_samples", R, N, 3), g("t", R, N)]
        ctx.mark_non_differentiable(*outs[9::10])
        if need_grad:
            ctx.save_for_backward(c, d, arenas[0], arenas[1], noise_c, noise_f, packed_c, packed_f, *saves)
            ctx.cfg, ctx.plan, ctx.npass, ctx.c2f_off = cfg, plan, len(passes), c2f_off
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g):
        lib = L.load()
        cfg, plan, npass = ctx.cfg, ctx.plan, ctx.npass
        c, d, arena_ray, arena_samp, noise_c, noise_f, packed_c, packed_f, *saves = ctx.saved_tensors
        dev = c.device
        R, Nc, Nf = cfg["R"], cfg["Nc"], cfg["Nf"]
        bases = (arena_ray.data_ptr(), arena_samp.data_ptr())
        A = lambda name: bases[plan.off[name][2]] + 4 * plan.off[name][0]
        pose = bool(ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        c2f_of = lambda tag: ctx.c2f_off if ctx.c2f_off is not None else A(tag + "c2f")       # the vector the forward of that pass was given
        passes = [("c", Nc, cfg["prec_c"], packed_c, noise_c, c2f_of("c"), saves[0], g[0:9])]
        if npass == 2:
            passes.append(("f", Nc + Nf, cfg["prec_f"], packed_f, noise_f, c2f_of("f"), saves[1], g[10:19]))
        active = [p for p in passes if any(x is not None for x in p[7])]
        gp = torch.empty(2, L.N_PARAMS, dtype=torch.float32, device=dev)
        rays = torch.empty(2, R, 3, dtype=torch.float32, device=dev) if pose else None
        stream = L.stream_ptr(dev)
        keep = []
        with L.on(dev):
            first = True
            for tag, N, prec, packed, noise, c2f, save, gs in active:
                gs = [(_contig32(x) if x is not None else None) for x in gs]
                ws = torch.empty(lib.sparf_bwd_workspace_bytes(prec, R, N, int(pose)), dtype=torch.uint8, device=dev)
                tables = L.tables_device(L.base_prec(prec), dev)
                P = lambda x: x.data_ptr() if x is not None else None
                a = L.PassBwd(prec=prec, nrays=R, nsamp=N, center=c.data_ptr(), dir=d.data_ptr(), t=A(tag + "t"), noise=P(noise),
                              noise_scale=float(cfg["noise_scale"]) if noise is not None else 0.0, white_bg=int(cfg["white_bg"]), packed=packed.data_ptr(),
                              c2f=c2f, tables=tables.data_ptr(), save=save.data_ptr(), raylen=A(tag + "raylen"), sigma_raw=A(tag + "sigma_raw"),
                              rgb_samples=A(tag + "rgb_samples"), weights=A(tag + "weights"), g_rgb=P(gs[0]), g_depth=P(gs[1]), g_opacity=P(gs[2]),
                              g_weights=P(gs[3]), ws=ws.data_ptr(), grad_params=gp.data_ptr() + (0 if tag == "c" else 4 * L.N_PARAMS),
                              d_center=P(rays[0]) if pose else None, d_dir=P(rays[1]) if pose else None, g_depth_var=P(gs[4]), g_rgb_var=P(gs[5]),
                              g_all_cumulated=P(gs[6]), g_density=P(gs[7]), g_rgb_samples=P(gs[8]), accumulate_rays=0 if first else 1)
                L.check(lib.sparf_pass_backward(ctypes.byref(a), stream), "sparf_pass_backward")
                keep += [ws, gs]
                first = False
        tags = {p[0] for p in active}
        g_c = gp[0] if "c" in tags else None
        g_f = gp[1] if ("f" in tags and npass == 2) else None
        if not active:
            rays = None
        return (rays[0] if (rays is not None and ctx.needs_input_grad[0]) else None, rays[1] if (rays is not None and ctx.needs_input_grad[1]) else None,
                None, None, None, None, None, None, None, None, None, None, None, None, g_c, g_f)


class FlatParams(torch.autograd.Function):
    """The 20 parameter tensors of a network as ONE flat tensor (NeRF.flat_params): forward = one `cat`, backward = the flat gradient cut
    into the parameters' shapes as views of itself -- one autograd node where `torch.cat([p.reshape(-1) ...])` is eleven."""

    @staticmethod
    def forward(ctx, *params):
        ctx.shapes = [tuple(p.shape) for p in params]
        ctx.set_materialize_grads(False)          # a network whose pass received no gradient keeps p.grad = None (as under per-parameter inputs)
        return torch.cat([p.reshape(-1) for p in params])

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * len(ctx.shapes)
        sizes = [math.prod(s) for s in ctx.shapes]
        parts = g.split_with_sizes(sizes)
        return tuple(x if len(s) == 1 else x.view(s) for x, s in zip(parts, ctx.shapes))


RENDER_KEYS = ("rgb", "depth", "opacity", "weights", "depth_var", "rgb_var", "all_cumulated", "density_samples", "rgb_samples", "t")


def render_fused(center, dirs, cfg, jitter, u_mid, noise_c, noise_f, range_dev, packed_c, packed_f, far_packed_c, far_packed_f, prog_c, prog_f, theta_c, theta_f):
    """-> (coarse dict, fine dict | None) with the reference's composite keys + 't' (flat ray axis)"""
    cfg = dict(cfg, grad=torch.is_grad_enabled())
    flat = RenderFn.apply(center, dirs, cfg, jitter, u_mid, noise_c, noise_f, range_dev, packed_c, packed_f, far_packed_c, far_packed_f, prog_c, prog_f,
                          theta_c, theta_f)
    coarse = dict(zip(RENDER_KEYS, flat[:10]))
    fine = dict(zip(RENDER_KEYS, flat[10:20])) if cfg["fine"] else None
    return coarse, fine
