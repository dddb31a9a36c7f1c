"""Does joint pose-NeRF REGISTRATION work on this renderer?  

SPARF's point is `joint_pose_nerf_trainer.py:382-406, 513-549`: noisy initial poses are pulled onto the scene by the photometric loss
plus the multi-view correspondence loss (`corres_loss.py:50-223`) while the two NeRFs train.  Rounds 3-5 compared ARITHMETIC under pose
refinement on a textureless analytic scene, photometric loss only -- where no trainer registers anything.  This tool runs the reference's
registration recipe, restated (the reference's sources cannot travel to the GPU box), on a TEXTURED synthetic scene with EXACT synthetic
correspondences, twice from the same start:

  oracle   oracle/nerf_oracle.py as fp32 PyTorch-ROCm ops on the GPU (rocBLAS + torch autograd: the reference's own arithmetic path,
           none of this repo's kernels), torch.optim.Adam + clip_grad_norm_
  hip      sparf_amd `Graph` (the product: HIP kernels behind the reference's API) + FusedAdam, in the default precision mode

Both see identical initial weights, initial pose noise, rays, matches and random draws.  What is restated from the reference:
  scene / data      three views of one scene with overlapping fields of view (DTU's 3-view split is a small-baseline rig), initial poses =
                    ground truth composed with se(3) noise N(0, 0.15^2) (`dtu/sparf.py:32-33`, camera.noise)
  pose model        learnable se(3) refinements composed onto the initial poses (`joint_pose_nerf_trainer.py:710-749`), Adam, lr 1e-3 ->
                    1e-4 exponentially (`default_config.py:297-301`)
  per iteration     `Graph.render` at random pixel indices of all views: MSE on rgb + rgb_fine (`base_losses.py:151-153`); a random ordered
                    view pair (i, j) with matches p_i <-> p_j: `render_image_at_specific_pose_and_rays(pixels=)` in both views
                    (`corres_loss.py:158-166`), depth and depth_fine re-projected through the CURRENT relative pose, Huber(delta 1) on the
                    pixel error in both directions, averaged over the four terms (`corres_loss.py:73-95, 183-221`), weight 10^-2 (the
                    reference: 10^-3 over 200 k iterations, `dtu/sparf.py:66`; with 3 000 iterations the networks fit the three images
                    around the wrong poses before 10^-3 has moved them -- measured on the oracle alone, CPU: 4.0 deg left after 1 500 steps
                    at 10^-3, 1.6 deg after 1 200 at 10^-2); clip_grad_norm 0.1 per network + Adam 5e-4 (`nerf_trainer.py:181-185`)
  c2f               BARF band weights swept by `progress.data.fill_` (`nerf_trainer.py:271-275`), window [0.1, 0.5] of this (short) run
  exact matches     p_j = project(X(p_i, z_i^GT)); kept where X is visible in j (GT depth of j at p_j agrees) -- what
                    `base_corres_loss.py:130-147` would read from a perfect matcher
  pose error        rotation of the RELATIVE poses to view 0 against ground truth (gauge-free: a global rigid motion of all cameras leaves
                    every loss unchanged), and camera-centre distance after a similarity alignment (`evaluate_camera_alignment`)

    python tests/tools/registration_run.py --steps 3000 --seeds 3 --out gpurun_out/r06_registration.json
    python tests/tools/registration_run.py --device cpu --trainers oracle --steps 200 --rays 256 --samples 16 16 --hw 60 80     (plumbing, CPU)

Test infrastructure: imports oracle/ (allowed under tests/); the product path never does."""
import argparse
import json
import math
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]

from oracle import nerf_oracle as O                                      # noqa: E402
from bench_workloads import compose, injected_rng, project, se3_exp, to44    # noqa: E402
from sparf_amd.config import default_opt                                 # noqa: E402


# ---------------------------------------------------------------------------------------------- the scene
class TexturedScene:
    """A unit sphere at the origin inside a sphere of radius R_OUT seen from within, both with a band-limited procedural texture
    (sums of products of sines at 2-9 cycles per unit: corners and blobs at every scale the c2f sweep opens up).  Every ray hits a
    surface at a finite depth, so exact colour, z-depth and cross-view matches exist for every pixel."""
    R_OUT = 5.0

    def __init__(self, device, B=3, H=240, W=320, f=330.0, radius=3.2, spread_deg=16.0):
        self.dev, self.B, self.H, self.W = device, B, H, W
        poses = []
        for b in range(B):
            az = math.radians(spread_deg * (b - (B - 1) / 2))
            el = math.radians(10.0 + 6.0 * ((b % 2) * 2 - 1) * (0 if B == 1 else 1))
            c = torch.tensor([radius * math.sin(az) * math.cos(el), -radius * math.sin(el), -radius * math.cos(az) * math.cos(el)])
            z = -c / c.norm()
            x = torch.linalg.cross(torch.tensor([0.0, 1.0, 0.0]), z)
            x = x / x.norm()
            y = torch.linalg.cross(z, x)
            Rw2c = torch.stack([x, y, z], dim=1).T
            poses.append(torch.cat([Rw2c, (-Rw2c @ c)[:, None]], dim=1))
        self.pose_gt = torch.stack(poses).to(device)
        self.intr = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]]).repeat(B, 1, 1).to(device)
        self.depth_range = (0.8, radius + self.R_OUT + 0.3)

    @staticmethod
    def texture(p, which):
        s = torch.sin
        if which == 0:       # the inner sphere
            a = s(7.0 * p[..., 0] + 0.3) * s(6.0 * p[..., 1] + 1.1) + 0.5 * s(13.0 * p[..., 2] + 2.0 * p[..., 0])
            b = s(5.0 * p[..., 1] - 0.7) * s(9.0 * p[..., 2] + 0.2) + 0.5 * s(11.0 * p[..., 0] - 3.0 * p[..., 1])
            c = s(8.0 * p[..., 2] + 1.9) * s(4.0 * p[..., 0] - 0.4) + 0.5 * s(15.0 * p[..., 1] + 1.0 * p[..., 2])
        else:                # the enclosing sphere (points at radius 5: lower spatial frequencies)
            a = s(1.7 * p[..., 0] + 0.9) * s(1.3 * p[..., 1] - 0.2) + 0.5 * s(2.9 * p[..., 2] + 0.7 * p[..., 1])
            b = s(1.1 * p[..., 1] + 2.3) * s(2.1 * p[..., 2] + 0.5) + 0.5 * s(3.1 * p[..., 0] - 0.9 * p[..., 2])
            c = s(1.9 * p[..., 2] - 1.1) * s(1.5 * p[..., 0] + 1.4) + 0.5 * s(2.7 * p[..., 1] + 0.6 * p[..., 0])
        return (0.5 + torch.stack([a, b, c], dim=-1) / 3.0).clamp(0.0, 1.0)

    def trace(self, pose, intr, px):
        """pose [3,4] w2c, intr [3,3], px [N,2] pixel coordinates (used as given) -> colour [N,3], z-depth [N], world point [N,3]"""
        hom = torch.cat([px, torch.ones_like(px[:, :1])], dim=-1) @ torch.linalg.inv(intr).T          # camera-frame ray with z = 1
        R, t = pose[:, :3], pose[:, 3]
        o = -(R.T @ t)
        d = hom @ R                                                                                  # R^T hom, world frame, unnormalised (z-depth parametrisation)
        a = (d * d).sum(-1)
        bq = (d * o).sum(-1)
        co = (o * o).sum()
        disc_in = bq * bq - a * (co - 1.0)
        hit_in = disc_in > 0
        z_in = (-bq - disc_in.clamp(min=0).sqrt()) / a
        hit_in = hit_in & (z_in > 0)
        z_out = (-bq + (bq * bq - a * (co - self.R_OUT ** 2)).sqrt()) / a                            # the camera is inside: one positive root
        z = torch.where(hit_in, z_in, z_out)
        X = o + d * z[:, None]
        col = torch.where(hit_in[:, None], self.texture(X, 0), self.texture(X, 1))
        return col, z, X

    def images(self):
        H, W = self.H, self.W
        ys, xs = torch.meshgrid(torch.arange(H, device=self.dev, dtype=torch.float32) + 0.5, torch.arange(W, device=self.dev, dtype=torch.float32) + 0.5, indexing="ij")
        px = torch.stack([xs, ys], dim=-1).reshape(-1, 2)
        return torch.stack([self.trace(self.pose_gt[b], self.intr[b], px)[0] for b in range(self.B)])      # [B, HW, 3]

    def matches(self, i, j, n, gen):
        """n exact matches p_i <-> p_j of the ordered view pair (i, j): the scene point behind p_i, seen in j and not occluded there"""
        px = torch.rand(3 * n, 2, generator=gen, device=self.dev) * torch.tensor([self.W - 1.0, self.H - 1.0], device=self.dev)
        _, z_i, _ = self.trace(self.pose_gt[i], self.intr[i], px)
        T = to44(self.pose_gt)
        uv, z_proj = project(px, z_i, self.intr[i], self.intr[j], T[j] @ torch.linalg.inv(T[i]))
        inside = (uv[:, 0] >= 0) & (uv[:, 1] >= 0) & (uv[:, 0] <= self.W - 1) & (uv[:, 1] <= self.H - 1) & (z_proj > 0)
        _, z_j, _ = self.trace(self.pose_gt[j], self.intr[j], uv)
        keep = inside & ((z_j - z_proj).abs() <= 2e-3 * z_proj)
        idx = keep.nonzero()[:n, 0]
        return px[idx], uv[idx]


# ---------------------------------------------------------------------------------------------- shared loss (restated)
def huber_mean(diff, delta=1.0):
    return torch.nn.functional.huber_loss(diff, torch.zeros_like(diff), reduction="none", delta=delta).mean()


def corres_loss(ret_i, ret_j, p_i, p_j, K_i, K_j, pose_i, pose_j):
    """corres_loss.py:183-221: both depths (coarse, fine) x both directions, Huber on the re-projection error in pixels, / 4"""
    T = to44(torch.stack([pose_i, pose_j]))
    T_ij = T[1] @ torch.linalg.inv(T[0])
    T_ji = torch.linalg.inv(T_ij)
    total, n = 0.0, 0
    for key in ("depth", "depth_fine"):
        if key not in ret_i:
            continue
        d_i, d_j = ret_i[key].reshape(-1), ret_j[key].reshape(-1)
        uv, _ = project(p_i, d_i, K_i, K_j, T_ij)
        vu, _ = project(p_j, d_j, K_j, K_i, T_ji)
        total = total + huber_mean(uv - p_j) + huber_mean(vu - p_i)
        n += 2
    return total / n


def relative_rotation_error_deg(pose, pose_gt):
    """mean angle between R_b R_0^T (estimated) and its ground truth, b = 1..B-1: invariant to a global rigid motion of all cameras"""
    R, Rg = pose[:, :, :3], pose_gt[:, :, :3]
    rel, relg = R[1:] @ R[:1].transpose(-1, -2), Rg[1:] @ Rg[:1].transpose(-1, -2)
    cos = ((rel @ relg.transpose(-1, -2)).diagonal(dim1=-2, dim2=-1).sum(-1) - 1) / 2
    return float(torch.rad2deg(torch.acos(cos.clamp(-1, 1))).mean())


def aligned_centre_error(pose, pose_gt):
    """camera centres after the similarity (Umeyama) that maps the estimated centres onto the true ones; mean distance"""
    c = -(pose[:, :, :3].transpose(-1, -2) @ pose[:, :, 3:])[..., 0].double()
    g = -(pose_gt[:, :, :3].transpose(-1, -2) @ pose_gt[:, :, 3:])[..., 0].double()
    mc, mg = c.mean(0), g.mean(0)
    C, G = c - mc, g - mg
    U, S, Vt = torch.linalg.svd(G.T @ C)
    D = torch.eye(3, dtype=torch.float64, device=c.device)
    D[2, 2] = torch.sign(torch.linalg.det(U @ Vt))
    Rm = U @ D @ Vt
    s = (S * D.diagonal()).sum() / (C * C).sum().clamp(min=1e-30)
    return float(((s * (C @ Rm.T) + mg) - g).norm(dim=-1).mean())


# ---------------------------------------------------------------------------------------------- the two trainers
class OracleSide:
    def __init__(self, opt, scene, pose_init, sd_c, sd_f, lr_pose):
        self.opt, self.scene = opt, scene
        self.pc = {k: v.detach().clone().requires_grad_(k != "progress") for k, v in sd_c.items()}
        self.pf = {k: v.detach().clone().requires_grad_(k != "progress") for k, v in sd_f.items()}
        self.groups = [[v for k, v in p.items() if k != "progress"] for p in (self.pc, self.pf)]
        self.optim = torch.optim.Adam(self.groups[0] + self.groups[1], lr=5e-4)
        self.init_pose = pose_init.clone()
        self.se3 = torch.zeros(scene.B, 6, device=scene.dev, requires_grad=True)
        self.optim_pose = torch.optim.Adam([self.se3], lr=lr_pose)

    def poses(self):
        return compose(se3_exp(self.se3), self.init_pose)

    def set_progress(self, p):
        for d in (self.pc, self.pf):
            d["progress"].fill_(p)

    def render_idx(self, poses, idx, draws, mode="train"):
        center, ray = O.rays_at_index(poses, self.scene.intr, self.scene.H, self.scene.W, idx)
        kw = dict(jitter=draws[0], grid=draws[1]) if draws is not None else {}
        return O.render(self.opt, self.pc, self.pf, center, ray, list(self.scene.depth_range), mode=mode, it=None, **kw)

    def render_px(self, pose, view, px, draws):
        center, ray = O.rays_at_pixels(pose[None], self.scene.intr[view:view + 1], px[None])
        return O.render(self.opt, self.pc, self.pf, center, ray, list(self.scene.depth_range), mode="train", it=None, jitter=draws[0], grid=draws[1])

    def zero_grad(self):
        self.optim.zero_grad(set_to_none=True)
        self.optim_pose.zero_grad(set_to_none=True)

    def photometric(self, ret, target):
        e1, e2 = (ret["rgb"] - target) ** 2, (ret["rgb_fine"] - target) ** 2
        return e1.sum() / (e1.nelement() + 1e-6) + e2.sum() / (e2.nelement() + 1e-6)

    def update(self, lr_pose):
        for g in self.groups:
            torch.nn.utils.clip_grad_norm_(g, 0.1)
        self.optim.step()
        self.optim_pose.param_groups[0]["lr"] = lr_pose
        self.optim_pose.step()


class HipSide:
    def __init__(self, opt, scene, pose_init, sd_c, sd_f, lr_pose):
        from bench_workloads import PoseGraph
        from sparf_amd.optim import FusedAdam
        self.opt, self.scene = opt, scene
        self.graph = PoseGraph(opt, scene.dev, pose_init)
        for net, sd in ((self.graph.nerf, sd_c), (self.graph.nerf_fine, sd_f)):
            net.load_state_dict({k: v.to(scene.dev) for k, v in sd.items()}, strict=True)
            net.weights_changed()
        self.optim = FusedAdam([self.graph.nerf, self.graph.nerf_fine], lr=5e-4, max_grad_norm=0.1)
        self.optim_pose = torch.optim.Adam([self.graph.se3_refine], lr=lr_pose)
        self.data = dict(depth_range=torch.tensor([list(scene.depth_range)] * scene.B, dtype=torch.float32, device=scene.dev))
        self.rng = torch.tensor(list(scene.depth_range), dtype=torch.float32, device=scene.dev)

    def poses(self):
        return self.graph.get_w2c_pose(self.opt, None, mode="train")

    def set_progress(self, p):
        self.graph.nerf.progress.data.fill_(p)
        self.graph.nerf_fine.progress.data.fill_(p)

    def render_idx(self, poses, idx, draws, mode="train"):
        s = self.scene
        with injected_rng(draws[0] if draws else None, draws[1] if draws else None, []):
            return self.graph.render(self.opt, poses, H=s.H, W=s.W, intr=s.intr, ray_idx=idx, depth_range=self.rng, iter=None, mode=mode)

    def render_px(self, pose, view, px, draws):
        s = self.scene
        from sparf_amd.edict import EasyDict as edict
        with injected_rng(draws[0], draws[1], []):
            return self.graph.render_image_at_specific_pose_and_rays(self.opt, edict(self.data), pose, s.intr[view], s.H, s.W, iter=None, pixels=px, mode="train")

    def zero_grad(self):
        self.optim.zero_grad(set_to_none=True)
        self.optim_pose.zero_grad(set_to_none=True)

    def photometric(self, ret, target):
        from sparf_amd import ops
        return ops.photometric_loss(ret.rgb, target, rgb_fine=ret.rgb_fine)

    def update(self, lr_pose):
        self.optim.step()
        self.optim_pose.param_groups[0]["lr"] = lr_pose
        self.optim_pose.step()


def make_opt(samples, precision, c2f=(0.1, 0.5)):
    nc, nf = samples
    return default_opt(nerf=dict(fine_sampling=True, sample_intvs=nc, sample_intvs_fine=nf, rand_rays=0, depth=dict(param="metric")),
                       barf_c2f=list(c2f), **(dict(hip=dict(precision=precision)) if precision else {}))


def run_seed(args, seed, dev, say):
    torch.backends.cuda.matmul.allow_tf32 = False
    scene = TexturedScene(dev, H=args.hw[0], W=args.hw[1], f=args.hw[1] * 1.03)
    images = scene.images()
    opt = make_opt(args.samples, args.precision, args.c2f)
    Nc, Nf = args.samples
    g0 = torch.Generator().manual_seed(1000 + seed)
    noise = (torch.randn(scene.B, 6, generator=g0) * args.pose_noise).to(dev)
    pose_init = compose(se3_exp(noise), scene.pose_gt)
    sd_c = {k: v.to(dev) for k, v in O.init_params(opt, 100 + seed).items()}
    sd_f = {k: v.to(dev) for k, v in O.init_params(opt, 200 + seed, fine=True).items()}
    sides = {}
    for name in args.trainers:
        sides[name] = (OracleSide if name == "oracle" else HipSide)(opt, scene, pose_init, sd_c, sd_f, args.lr_pose)
    gen = torch.Generator(device=dev).manual_seed(77 + seed)
    R = args.rays // 2 // scene.B                 # photometric rays per view; the other half of the budget: matches of the view pair
    n_match = args.rays // 4
    held = torch.randperm(scene.H * scene.W, generator=gen, device=dev)[:1024]

    def evaluate(step):
        row = dict(step=step)
        with torch.no_grad():
            for name, s in sides.items():
                p = s.poses().detach()
                out = s.render_idx(p, held, None, mode="val")
                mse = ((out["rgb_fine"] - images[:, held]) ** 2).mean()
                row[name] = dict(rot_err_deg=relative_rotation_error_deg(p, scene.pose_gt), centre_err=aligned_centre_error(p, scene.pose_gt),
                                 psnr_train_views=float(-10 * torch.log10(mse)))
        return row

    curve = [evaluate(0)]
    say(json.dumps(dict(seed=seed, **curve[-1])))
    t0 = time.perf_counter()
    times = {k: 0.0 for k in sides}
    for it in range(args.steps):
        idx = torch.randperm(scene.H * scene.W, generator=gen, device=dev)[:R]
        i = int(torch.randint(0, scene.B, (1,), generator=gen, device=dev))
        j = (i + 1 + int(torch.randint(0, scene.B - 1, (1,), generator=gen, device=dev))) % scene.B
        p_i, p_j = scene.matches(i, j, n_match, gen)
        n = p_i.shape[0]
        draws = [(torch.rand(scene.B, R, Nc, 1, generator=gen, device=dev), torch.rand(Nf + 1, generator=gen, device=dev))] + \
                [(torch.rand(1, n, Nc, 1, generator=gen, device=dev), torch.rand(Nf + 1, generator=gen, device=dev)) for _ in range(2)]
        target = images[:, idx]
        lr_pose = args.lr_pose * (args.lr_pose_end / args.lr_pose) ** (it / max(1, args.steps - 1))
        for name, s in sides.items():
            if dev.type == "cuda":
                torch.cuda.synchronize()
            t1 = time.perf_counter()
            s.set_progress(it / args.steps)
            s.zero_grad()
            poses = s.poses()
            ret = s.render_idx(poses, idx, draws[0])
            ret_i = s.render_px(poses[i], i, p_i, draws[1])
            ret_j = s.render_px(poses[j], j, p_j, draws[2])
            loss = s.photometric(ret, target) + args.w_corres * corres_loss(ret_i, ret_j, p_i, p_j, scene.intr[i], scene.intr[j], poses[i], poses[j])
            loss.backward()
            s.update(lr_pose)
            if dev.type == "cuda":
                torch.cuda.synchronize()
            times[name] += time.perf_counter() - t1
        if (it + 1) % args.eval_every == 0 or it + 1 == args.steps:
            curve.append(evaluate(it + 1))
            say(json.dumps(dict(seed=seed, seconds=round(time.perf_counter() - t0, 1), **curve[-1])))
        if time.perf_counter() - t0 > args.max_seconds:
            break
    return dict(seed=seed, initial_pose_noise=args.pose_noise, curve=curve, ms_per_iteration={k: v / max(1, curve[-1]["step"]) * 1e3 for k, v in times.items()})


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--rays", type=int, default=4096, help="rays per iteration: half photometric (all views), half the two match renders")
    ap.add_argument("--samples", type=int, nargs=2, default=[64, 128])
    ap.add_argument("--hw", type=int, nargs=2, default=[240, 320])
    ap.add_argument("--pose-noise", type=float, default=0.15, help="std of the se(3) noise on the initial poses (dtu/sparf.py: camera.noise)")
    ap.add_argument("--lr-pose", type=float, default=1e-3)
    ap.add_argument("--lr-pose-end", type=float, default=1e-4)
    ap.add_argument("--w-corres", type=float, default=1e-2, help="weight of the correspondence loss (the reference: 10^-3 over 200 k iterations, dtu/sparf.py:66; these runs have a few thousand)")
    ap.add_argument("--c2f", type=float, nargs=2, default=[0.1, 0.5], help="BARF c2f window as fractions of the run (the reference: [0.4, 0.7] of 200 k iterations)")
    ap.add_argument("--trainers", nargs="*", default=["oracle", "hip"])
    ap.add_argument("--precision", default=None, help="HIP precision mode (default: the product's)")
    ap.add_argument("--eval-every", type=int, default=250)
    ap.add_argument("--max-seconds", type=float, default=1e9, help="per seed")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--out", default=None)
    ap.add_argument("--quiet", action="store_true")
    return ap.parse_args(argv)


def run(args):
    dev = torch.device(args.device)
    say = (lambda *a, **k: None) if args.quiet else (lambda *a, **k: print(*a, flush=True, **k))
    runs = [run_seed(args, seed, dev, say) for seed in range(args.seeds)]
    final = {}
    for name in args.trainers:
        e0 = [r["curve"][0][name]["rot_err_deg"] for r in runs]
        e1 = [r["curve"][-1][name]["rot_err_deg"] for r in runs]
        final[name] = dict(rot_err_deg_initial=e0, rot_err_deg_final=e1, centre_err_final=[r["curve"][-1][name]["centre_err"] for r in runs],
                           psnr_final=[r["curve"][-1][name]["psnr_train_views"] for r in runs])
    if len(args.trainers) == 2:
        a, b = args.trainers
        final["paired_rot_err_delta_deg"] = [y - x for x, y in zip(final[a]["rot_err_deg_final"], final[b]["rot_err_deg_final"])]
    doc = dict(what="joint pose-NeRF registration (photometric + correspondence loss, restated from the reference) on a textured synthetic scene with exact matches: "
                    "oracle (fp32 torch ops) and HIP renderer side by side from identical starts",
               steps=args.steps, rays_per_iteration=args.rays, samples=list(args.samples), image_hw=list(args.hw), trainers=args.trainers,
               precision=args.precision or "default", final=final, runs=runs, torch=torch.__version__)
    say(json.dumps(final))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(doc, f, indent=1)
    return doc


if __name__ == "__main__":
    run(parse())
