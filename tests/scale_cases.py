"""Benchmark-scale parity cases: one full `Graph` render + backward at the shapes of BASELINE.json
configs 1-4, refereed by the oracle in float64 (oracle.pass_fixed, referee mode).

Shared by tests/test_00_scale_gpu.py (asserts the bounds) and tests/tools/scale_parity.py (writes the
measured numbers to profiles/, where bench.py picks them up for its `parity` field).

Protocol (per config and precision mode):
  1. GPU, end to end through the public API: Graph.render / the loss-facing wrappers with the
     random draws injected (jitter, fine grid, sigma noise), loss = fixed random linear functional
     of rgb / depth / opacity / weights (+ _fine), backward.  origins / viewdirs keep their
     gradient (retain_grad) when poses are trained.
  2. Referee: the oracle on the CPU, float64 downstream of the fp32 encoding arguments, on the
     SAME rays, coarse depths, merged fine depths and noise (all taken from step 1: they are the
     inputs of the two MLP passes; what produced them is checked separately in 3), in ray chunks.
     Same loss, backward: parameter gradients of both networks, d loss / d origins, d viewdirs,
     and d pose by pushing the referee's ray gradients through the oracle's float64 ray generation.
  3. Feeder checks at scale: coarse depths bit-exact against the oracle's fp32 sample_depth; the
     merged fine depths against oracle.sample_pdf evaluated on the GPU's own coarse weights.
Errors: outputs max|a-b| / max|b| per tensor (the north_star's 1e-4 bar); gradients relative L2
per tensor (worst tensor reported) and max-norm relative.
"""
import contextlib
import math
import time

import numpy as np
import torch

from oracle import nerf_oracle as O
from sparf_amd.config import default_opt, _merge
from sparf_amd.edict import EasyDict as edict
from sparf_amd.renderer import Graph
from tests.golden.recipe import make_state_dict, ring_cameras

OUT_KEYS = ("rgb", "depth", "opacity", "weights", "depth_var", "all_cumulated", "rgb_samples", "density_samples")
LOSS_KEYS = ("rgb", "depth", "opacity", "weights")
# what the trainers / losses read from a render (all_cumulated only from render_to_max) ...
RENDERED = ("rgb", "depth", "opacity", "weights", "depth_var")
# ... and what is returned but never consumed downstream
PER_SAMPLE = ("rgb_samples", "density_samples", "all_cumulated")


def worst_of(outputs, keys):
    return max(v for k, v in outputs.items() if k.replace("_fine", "") in keys)

# BASELINE.json configs[1..4] (SURVEY.md section 8 config matrix): image size, views x rays, depth
# parametrisation and the option overrides of the reference settings file each one names.
CONFIGS = {
    1: dict(name="configs[1] DTU scan65-shaped, GT poses, 4x1024 rays (nerf_training_w_gt_poses/dtu/nerf.py)",
            B=4, R=1024, H=300, W=400, f=500.0, rng=[1.2, 5.2], sel="idx", pose_grad=False, progress=None, iter=10000,
            over=dict(nerf=dict(density_noise_reg=True))),
    2: dict(name="configs[2] DTU joint pose-NeRF, BARF c2f [0.4,0.7], 3x1365 rays, pose gradients (joint_pose_nerf_training/dtu/barf.py)",
            B=3, R=1365, H=300, W=400, f=500.0, rng=[1.2 * 0.8, 5.2 * 1.2], sel="idx", pose_grad=True, progress=0.55, iter=10000,
            over=dict(barf_c2f=[0.4, 0.7])),
    3: dict(name="configs[3] LLFF fern-shaped 378x504, inverse depth [1,0], pixel-path wrapper + render_up_to_maxdepth (llff/sparf.py)",
            B=1, R=2048, H=378, W=504, f=420.0, rng=[1, 0], sel="pixels", pose_grad=True, progress=None, iter=10000, to_max=True,
            over=dict(nerf=dict(depth=dict(param="inverse", range=[1, 0]), density_noise_reg=False))),
    4: dict(name="configs[4] Replica room0-shaped 340x600, 9x455 rays, fine gated until 25% (replica/sparf.py)",
            B=9, R=455, H=340, W=600, f=300.0, rng=[0.1, 6.5], sel="idx", pose_grad=True, progress=None, iter=60000,
            over=dict(max_iter=200000, nerf=dict(ratio_start_fine_sampling_at_x=0.25, density_noise_reg=False))),
}


def case_opt(cfg, precision, nc=64, nf=128):
    """precision "bf16x3" = the default: inverse-depth passes route the last samples of every ray to fp32
    (frequency_nerf.pass_precision); "bf16x3!" = bf16x3 kept on every row (opt.hip.inverse_depth_precision = 'bf16x3');
    "bf16x3#" = inverse-depth passes on the fp32 kernels as a whole (= 'fp32', round 3's behaviour)"""
    o = default_opt(nerf=dict(fine_sampling=True, sample_intvs=nc, sample_intvs_fine=nf, rand_rays=cfg["B"] * cfg["R"],
                              depth=dict(param="metric")))
    _merge(o, cfg["over"])
    how = dict(inverse_depth_precision="bf16x3") if precision.endswith("!") else dict(inverse_depth_precision="fp32") if precision.endswith("#") else {}
    _merge(o, dict(hip=dict(precision=precision.rstrip("!#"), **how)))
    return o


@contextlib.contextmanager
def injected_rng(jitter, grid, noises):
    """torch.rand / torch.randn of the renderer return the given draws (renderer.py:406, :439,
    frequency_nerf.py:192), in call order."""
    real_rand, real_randn = torch.rand, torch.randn
    noises = list(noises)

    def rand(*size, **kw):
        if len(size) == 4 and jitter is not None and tuple(size) == tuple(jitter.shape):
            return jitter.to(kw.get("device", "cpu"))
        if len(size) == 1 and grid is not None and size[0] == grid.numel():
            return grid.clone()
        return real_rand(*size, **kw)

    def randn(*size, **kw):
        if noises:
            n = noises.pop(0)
            assert int(np.prod(size)) == n.numel(), (size, n.shape)
            return n.reshape(*size).to(kw.get("device", "cpu"))
        return real_randn(*size, **kw)

    torch.rand, torch.randn = rand, randn
    try:
        yield
    finally:
        torch.rand, torch.randn = real_rand, real_randn


def max_rel(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def rel_l2(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-300))


def _loss_weights(rs, shapes):
    """fixed random linear functional; each term scaled so that it contributes O(1)"""
    return {k: torch.from_numpy((rs.uniform(-1, 1, size=s) / math.sqrt(max(1, int(np.prod(s))))).astype(np.float32)) for k, s in shapes.items()}


def referee(opt, sd_c, sd_f, center, ray, t, t_fine, noise_c, noise_f, lw, mode, chunk=512, dtype=torch.float64, want_ray_grad=True,
            device="cpu", loss_fn=None):
    """Oracle (`dtype` downstream of the fp32 encoding arguments) on fixed rays / depths, in ray
    chunks; returns outputs, parameter gradients (dicts per network) and ray gradients (on the CPU).
    center, ray [1,N,3]; t [1,N,Nc,1]; t_fine [1,N,Nt,1] or None; lw[key] [1,N,...].
    device: where the oracle's PyTorch ops execute.  "cpu" is the oracle as pinned; a cuda device
    runs the very same float64 PyTorch code through PyTorch-ROCm's own kernels (rocBLAS fp64 GEMMs,
    none of this repo's HIP code) -- seconds instead of minutes at 4096 rays, used by the test suite;
    tests/tools/scale_parity.py --referee-device cpu measured the same numbers on the CPU.
    loss_fn(part, ray_slice) -> scalar: a non-linear loss on the chunk's outputs (the photometric MSE of
    tests/tools/psnr_curve.py) instead of the linear functional `lw` (pass lw = {"rgb": <anything>} to enable gradients)."""
    cd = None if dtype == torch.float32 else dtype
    rdev = torch.device(device)
    _cpu = lambda x: x.detach().to(rdev) if x is not None else None
    center, ray, t, t_fine, noise_c, noise_f = (_cpu(x) for x in (center, ray, t, t_fine, noise_c, noise_f))
    lw = {k: (v.to(rdev) if torch.is_tensor(v) else v) for k, v in lw.items()}
    pc = {k: v.detach().to(rdev, dtype).requires_grad_(k != "progress" and bool(lw)) for k, v in sd_c.items()}
    pf = {k: v.detach().to(rdev, dtype).requires_grad_(k != "progress" and bool(lw)) for k, v in sd_f.items()} if t_fine is not None else None
    N = ray.shape[1]
    outs, d_c, d_r = [], [], []
    torch.set_grad_enabled(bool(lw))
    for i in range(0, N, chunk):
        s = slice(i, min(i + chunk, N))
        c = center[:, s].clone().requires_grad_(want_ray_grad and bool(lw))
        r = ray[:, s].clone().requires_grad_(want_ray_grad and bool(lw))
        o = O.pass_fixed(opt, pc, c, r, t[:, s], mode=mode, noise=noise_c[:, s] if noise_c is not None else None, compute_dtype=cd)
        part = {k: o[k] for k in OUT_KEYS}
        if t_fine is not None:
            of = O.pass_fixed(opt, pf, c, r, t_fine[:, s], mode=mode, noise=noise_f[:, s] if noise_f is not None else None,
                              fine=True, compute_dtype=cd)
            part.update({k + "_fine": of[k] for k in OUT_KEYS})
        loss = loss_fn(part, s) if loss_fn is not None else sum((part[k] * lw[k][:, s].to(part[k].dtype)).sum() for k in lw if k in part)
        if lw:
            loss.backward()
        outs.append({k: v.detach().cpu() for k, v in part.items()})
        if want_ray_grad and lw:
            d_c.append(c.grad.double().cpu())
            d_r.append(r.grad.double().cpu())
    torch.set_grad_enabled(True)
    out = {k: torch.cat([p[k] for p in outs], dim=1) for k in outs[0]}
    grads = dict(nerf={k: v.grad.cpu() for k, v in pc.items() if k != "progress" and v.grad is not None},
                 nerf_fine={k: v.grad.cpu() for k, v in pf.items() if k != "progress" and v.grad is not None} if pf is not None else {})
    return out, grads, (torch.cat(d_c, 1) if d_c else None), (torch.cat(d_r, 1) if d_r else None)


def run_case(cfg_id, precision, device=None, yardstick=False, chunk=512, seed=0, nc=64, nf=128, rays_scale=1.0, referee_device="cpu", log=print):
    """Returns a dict: output errors, gradient errors (worst tensor), feeder checks, timings.
    rays_scale < 1 shrinks the ray count (quick CPU-side plumbing checks of this module)."""
    cfg = dict(CONFIGS[cfg_id])
    cfg["R"] = max(1, int(round(cfg["R"] * rays_scale)))
    dev = torch.device(device or "cuda:0")
    opt = case_opt(cfg, precision, nc, nf)
    B, R, H, W = cfg["B"], cfg["R"], cfg["H"], cfg["W"]
    Nt = nc + nf
    graph = Graph(opt, dev)
    graph.nerf.load_state_dict(make_state_dict(opt, 100 + cfg_id, cfg["progress"]))
    graph.nerf_fine.load_state_dict(make_state_dict(opt, 200 + cfg_id, cfg["progress"]))
    pose, intr = ring_cameras(B, seed=cfg_id, H=H, W=W, f=cfg["f"])
    rs = np.random.RandomState(1000 * cfg_id + seed)
    if cfg["sel"] == "pixels":
        pixels = torch.from_numpy(rs.uniform(0, [W - 1, H - 1], size=(R, 2)).astype(np.float32))       # float (x, y), no +0.5 (camera.py:400-406)
        sel = dict(pixels=pixels.to(dev))
    else:
        ray_idx = torch.from_numpy(rs.permutation(H * W)[:R])
        sel = dict(ray_idx=ray_idx.to(dev))
    train_noise = bool(opt.nerf.density_noise_reg)
    jitter = torch.from_numpy(rs.uniform(size=(B, R, nc, 1)).astype(np.float32))
    grid = torch.from_numpy(rs.uniform(size=nf + 1).astype(np.float32))
    noise_c = torch.from_numpy(rs.normal(size=(B * R, nc)).astype(np.float32)) if train_noise else None
    noise_f = torch.from_numpy(rs.normal(size=(B * R, Nt)).astype(np.float32)) if train_noise else None
    shapes = {"rgb": (B, R, 3), "depth": (B, R, 1), "opacity": (B, R, 1), "weights": (B, R, nc, 1),
              "rgb_fine": (B, R, 3), "depth_fine": (B, R, 1), "opacity_fine": (B, R, 1), "weights_fine": (B, R, Nt, 1)}
    lw = _loss_weights(rs, shapes)

    # ---- 1. GPU, end to end through the public API
    pg = pose.to(dev).requires_grad_(cfg["pose_grad"])
    data = edict(idx=torch.arange(B), image=torch.zeros(B, 3, H, W, device=dev), intr=intr.to(dev), pose=pg,
                 depth_range=torch.tensor([cfg["rng"]] * B, dtype=torch.float32, device=dev))
    t0 = time.perf_counter()
    with injected_rng(jitter, grid, [n for n in (noise_c, noise_f) if n is not None]):
        if cfg["sel"] == "pixels":      # the correspondence / depth-consistency call shape (corres_loss.py:158-166): one (3,4) pose, (N,2) pixels
            ret = graph.render_image_at_specific_pose_and_rays(opt, data, pg[0], intr[0].to(dev), H, W, iter=cfg["iter"], mode="train", **sel)
        else:
            ret = graph.render(opt, pg, H=H, W=W, intr=intr.to(dev), depth_range=data.depth_range[0] if opt.nerf.depth.param == "metric" else cfg["rng"],
                               iter=cfg["iter"], mode="train", **sel)
    assert "rgb_fine" in ret, "fine pass expected at this iteration"
    if cfg["pose_grad"]:
        ret.origins.retain_grad()
        ret.viewdirs.retain_grad()
    loss = sum((ret[k] * lw[k].to(dev)).sum() for k in lw)
    graph.zero_grad(set_to_none=True)
    loss.backward()
    torch.cuda.synchronize(dev)
    t_gpu = time.perf_counter() - t0

    flat = lambda x: x.detach().reshape(1, B * R, *x.shape[2:])
    center, ray = flat(ret.origins).cpu(), flat(ret.viewdirs).cpu()
    t, t_fine = flat(ret.t).cpu(), flat(ret.t_fine).cpu()

    # ---- 3. feeder checks
    res = dict(config=cfg_id, name=cfg["name"], precision=precision, rays=B * R, rows_coarse=B * R * nc, rows_fine=B * R * Nt)
    rng = cfg["rng"]
    if opt.nerf.depth.param == "metric":        # device-tensor range: fp32 arithmetic (renderer._range)
        rng32 = torch.tensor(rng, dtype=torch.float32)
        t_ref = O.sample_depth(opt, B, R, nc, [rng32[0], rng32[1]], "train", jitter)
    else:
        t_ref = O.sample_depth(opt, B, R, nc, rng, "train", jitter)
    res["t_coarse_bit_exact"] = bool(torch.equal(ret.t.detach().cpu(), t_ref))
    res["t_coarse_maxrel"] = max_rel(ret.t, t_ref)
    with torch.no_grad():
        rngf = [float(torch.tensor(rng[0], dtype=torch.float32)), float(torch.tensor(rng[1], dtype=torch.float32))]
        tf_ref = O.sample_pdf(ret.weights.detach().cpu()[..., 0], nc, nf, rngf, grid)
        merged_ref = torch.cat([ret.t.detach().cpu(), tf_ref], dim=2).sort(dim=2).values
    res["t_fine_vs_sampler_oracle_maxabs"] = float((ret.t_fine.detach().cpu() - merged_ref).abs().max())
    res["t_fine_sorted"] = bool((ret.t_fine[:, :, 1:] >= ret.t_fine[:, :, :-1]).all())

    # ---- 2. referee
    sd_c, sd_f = graph.nerf.state_dict(), graph.nerf_fine.state_dict()
    lwf = {k: v.reshape(1, B * R, *v.shape[2:]) for k, v in lw.items()}
    ncf = noise_c.reshape(1, B * R, nc) if noise_c is not None else None
    nff = noise_f.reshape(1, B * R, Nt) if noise_f is not None else None
    t0 = time.perf_counter()
    ref, gref, dc_ref, dr_ref = referee(opt, sd_c, sd_f, center, ray, t, t_fine, ncf, nff, lwf, "train", chunk=chunk,
                                        want_ray_grad=cfg["pose_grad"], device=referee_device)
    res["referee_device"] = str(referee_device)
    res["referee_seconds"] = round(time.perf_counter() - t0, 1)
    res["gpu_seconds_first_call"] = round(t_gpu, 2)

    def compare(tag, got_out, got_grads, got_dc, got_dr, got_dpose, ref_out=ref):
        e = {}
        outs = {}
        for k in ref_out:
            if k not in got_out:
                continue
            g = got_out[k]
            g = g.reshape(1, B * R, *g.shape[2:]) if g.dim() >= 2 and g.shape[0] == B else g
            if k.startswith("rgb_var"):
                continue
            outs[k] = max_rel(g.reshape(ref_out[k].shape), ref_out[k])
        e["outputs"] = outs
        e["outputs_worst"] = max(outs.values())
        e["rendered_worst"] = worst_of(outs, RENDERED)
        e["per_sample_worst"] = worst_of(outs, PER_SAMPLE)
        gl2, gmx = {}, {}
        for net in ("nerf", "nerf_fine"):
            for k, gr in gref[net].items():
                gl2[f"{net}.{k}"] = rel_l2(got_grads[net][k], gr)
                gmx[f"{net}.{k}"] = max_rel(got_grads[net][k], gr)
        e["param_grad_rel_l2"] = gl2
        e["param_grad_rel_l2_worst"] = max(gl2.values())
        e["param_grad_maxrel_worst"] = max(gmx.values())
        allg = torch.cat([got_grads[n][k].detach().double().cpu().reshape(-1) for n in ("nerf", "nerf_fine") for k in gref[n]])
        allr = torch.cat([gref[n][k].double().reshape(-1) for n in ("nerf", "nerf_fine") for k in gref[n]])
        e["param_grad_rel_l2_all"] = float((allg - allr).norm() / allr.norm())
        if got_dc is not None:
            e["d_origins_rel_l2"] = rel_l2(got_dc.reshape(dc_ref.shape), dc_ref)
            e["d_viewdirs_rel_l2"] = rel_l2(got_dr.reshape(dr_ref.shape), dr_ref)
        if got_dpose is not None:
            e["d_pose_maxrel"] = max_rel(got_dpose, dpose_ref)
        return e

    dpose_ref = None
    if cfg["pose_grad"]:
        p64 = pose.double().requires_grad_(True)
        if cfg["sel"] == "pixels":
            c64, r64 = O.rays_at_pixels(p64[:1], intr.double()[:1], pixels.double()[None])
        else:
            c64, r64 = O.rays_at_index(p64, intr.double(), H, W, ray_idx)
        gc, gr = dc_ref.reshape(c64.shape), dr_ref.reshape(r64.shape)
        dpose_ref, = torch.autograd.grad([c64, r64], [p64], grad_outputs=[gc, gr])
        if cfg["sel"] == "pixels":
            dpose_ref = dpose_ref[:1]
    got_grads = {n: {k: p.grad for k, p in getattr(graph, n).named_parameters() if k != "progress"} for n in ("nerf", "nerf_fine")}
    got_dpose = pg.grad[:1] if (cfg["pose_grad"] and cfg["sel"] == "pixels") else pg.grad
    res["hip"] = compare("hip", ret, got_grads, ret.origins.grad if cfg["pose_grad"] else None,
                         ret.viewdirs.grad if cfg["pose_grad"] else None, got_dpose)
    res["loss"] = float(loss.detach())

    # ---- yardstick: the fp32 reference's own distance to the referee on the same inputs
    if yardstick:
        t0 = time.perf_counter()
        y, gy, dcy, dry = referee(opt, sd_c, sd_f, center, ray, t, t_fine, ncf, nff, lwf, "train", chunk=chunk, dtype=torch.float32,
                                  want_ray_grad=cfg["pose_grad"], device=referee_device)
        res["reference_fp32"] = compare("ref32", y, gy, dcy, dry, None)
        res["yardstick_seconds"] = round(time.perf_counter() - t0, 1)

    # ---- render_up_to_maxdepth under no_grad (depth_cons_loss.py:266-273): all_cumulated(_fine) vs the referee
    if cfg.get("to_max"):
        dmax = torch.from_numpy(rs.uniform(1.5, 6.0, size=(R,)).astype(np.float32))
        with torch.no_grad():
            rm = graph.render_up_to_maxdepth_at_specific_pose_and_rays(opt, data, pg[0].detach(), intr[0].to(dev), H, W, depth_max=dmax.to(dev),
                                                                       iter=cfg["iter"], mode="train", **sel)
        assert not rm.all_cumulated_fine.requires_grad
        tm = O.sample_depth_to_max(nc, float(cfg["rng"][0]), dmax[None])
        res["to_max_t_bit_exact"] = bool(torch.equal(rm.t.cpu(), tm))
        rmo, _, _, _ = referee(opt, sd_c, sd_f, flat(rm.origins).cpu(), flat(rm.viewdirs).cpu(), tm, tm, None, None, {}, "train",
                               chunk=chunk, want_ray_grad=False, device=referee_device)
        res["to_max"] = {k: max_rel(rm[k].reshape(rmo[k].shape), rmo[k]) for k in ("all_cumulated", "all_cumulated_fine", "depth", "rgb_fine")}
    del graph
    torch.cuda.empty_cache()
    return res
