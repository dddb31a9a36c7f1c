"""Side-by-side training runs: HIP bf16x3 / bf16 / fp32 against the fp32 oracle, ALL ON THE GPU.

    python tests/tools/psnr_curve.py --config 1 --steps 2000 --out gpurun_out/r03_psnr_curve_c1.json
    python tests/tools/psnr_curve.py --config 2 --steps 2000 --out gpurun_out/r03_psnr_curve_c2.json

BASELINE's second metric ("PSNR vs ref") at the benchmark's own batch: 4096 rays x (64 + 128) samples per
step.  Every trainer starts from the identical initialisation and sees the identical rays and random draws
(stratified jitter, the shared fine grid, both density-noise tensors: drawn once per step on the device and
injected into the renderer's torch.rand / torch.randn calls); what differs is the arithmetic:

  oracle   oracle/nerf_oracle.py as fp32 PyTorch-ROCm ops on cuda:0 (rocBLAS GEMMs, torch autograd -- the
           reference's own path, none of this repo's kernels), torch.optim.Adam + clip_grad_norm_(0.1) per network
  hip:*    sparf_amd.Graph + fused photometric loss + FusedAdam in the given precision mode

config 1: BASELINE configs[1] shape (4 views 300x400, fixed GT poses, density-noise regularisation).
config 2: configs[2] shape (3 noisy views, SE(3) refinement parameters behind get_w2c_pose, BARF c2f
          [0.4, 0.7] swept over the run: opt.max_iter = --steps).
config 3: configs[3] shape (LLFF-like forward-facing rig, 378x504, INVERSE depth [1, 0] -- samples out to t ~ 1e8, the last ones
          of every ray through the fp32 kernels in bf16x3 mode -- pose refinement + c2f as config 2), photometric loss only.
Held-out PSNR (-10 log10 MSE of rgb_fine, nerf_trainer.py:298-305 / metrics.py:246) of every trainer on the
same fixed rays every --eval-every steps, the final gap to the oracle, the pose error (config 2), and -- at
step 0 and --grad-check-at -- the parameter-gradient error of every HIP mode UNDER THE PHOTOMETRIC LOSS against
the float64 referee on the same rays / depths / draws (tests/scale_cases.referee), next to the fp32 oracle's own.

Test infrastructure: imports oracle/ (allowed under tests/); the product path never does.
"""
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
from bench_workloads import Workload, compose, injected_rng, se3_exp    # noqa: E402
from sparf_amd import ops                                                # noqa: E402
from tests.scale_cases import referee, rel_l2                            # noqa: E402


def psnr_of(pred, tgt):
    return float(-10.0 * torch.log10(((pred - tgt) ** 2).mean()))


def pose_error(pose, pose_gt):
    """mean rotation angle (degrees) and camera-centre distance between [B,3,4] w2c poses"""
    R, Rg = pose[:, :, :3], pose_gt[:, :, :3]
    cos = ((R @ Rg.transpose(-1, -2)).diagonal(dim1=-2, dim2=-1).sum(-1) - 1) / 2
    ang = torch.rad2deg(torch.acos(cos.clamp(-1, 1))).mean()
    c = -(R.transpose(-1, -2) @ pose[:, :, 3:])[..., 0]
    cg = -(Rg.transpose(-1, -2) @ pose_gt[:, :, 3:])[..., 0]
    return float(ang), float((c - cg).norm(dim=-1).mean())


class OracleTrainer:
    """the reference's PyTorch path (oracle restatement) trained with torch's own optimiser, on `device`"""

    def __init__(self, w, device):
        g = w.graph
        self.opt, self.dev = w.opt, device
        self.pc = {k: v.detach().clone().requires_grad_(k != "progress") for k, v in g.nerf.state_dict().items()}
        self.pf = {k: v.detach().clone().requires_grad_(k != "progress") for k, v in g.nerf_fine.state_dict().items()}
        self.groups = [[v for k, v in p.items() if k != "progress"] for p in (self.pc, self.pf)]
        self.optim = torch.optim.Adam(self.groups[0] + self.groups[1], lr=5e-4)
        self.pose_cfg = w.config != 1
        if self.pose_cfg:
            self.init_pose = g.init_pose.detach().clone()
            self.se3 = torch.zeros_like(g.se3_refine.detach()).requires_grad_(True)
            self.optim_pose = torch.optim.Adam([self.se3], lr=1e-3)
        self.pose_fixed, self.intr, self.H, self.W = w.data.pose, w.intr, w.H, w.W

    def poses(self):
        return compose(se3_exp(self.se3), self.init_pose) if self.pose_cfg else self.pose_fixed

    def set_progress(self, p):
        for d in (self.pc, self.pf):
            d["progress"].fill_(p)

    def render(self, idx, rng, mode, it, draws=None):
        center, ray = O.rays_at_index(self.poses(), self.intr, self.H, self.W, idx)
        kw = dict(jitter=draws[0], grid=draws[1], noise_c=draws[2], noise_f=draws[3]) if draws is not None else {}
        return O.render(self.opt, self.pc, self.pf, center, ray, rng, mode=mode, it=it, **kw)

    def step(self, idx, rng, it, draws, target):
        self.optim.zero_grad(set_to_none=True)
        if self.pose_cfg:
            self.optim_pose.zero_grad(set_to_none=True)
        out = self.render(idx, rng, "train", it, draws)
        e1, e2 = (out["rgb"] - target) ** 2, (out["rgb_fine"] - target) ** 2
        loss = e1.sum() / (e1.nelement() + 1e-6) + e2.sum() / (e2.nelement() + 1e-6)          # base_losses.py:151-153
        loss.backward()
        for gpar in self.groups:
            torch.nn.utils.clip_grad_norm_(gpar, 0.1)                                        # nerf_trainer.py:181-185
        self.optim.step()
        if self.pose_cfg:
            self.optim_pose.step()
        return loss.detach()


class HipTrainer:
    def __init__(self, config, precision, device, rays, steps, seed=0):
        self.w = w = Workload(config, precision, device, rays=rays, seed=7 + seed)
        if config != 1:
            w.opt.max_iter = w.max_iter = steps          # c2f progress sweeps 0 -> 1 over the run
        self.precision = precision

    def poses(self):
        w = self.w
        return w.graph.get_w2c_pose(w.opt, w.data, mode="train") if w.config != 1 else w.data.pose

    def set_progress(self, p):
        g = self.w.graph
        g.nerf.progress.data.fill_(p)
        g.nerf_fine.progress.data.fill_(p)

    def render(self, idx, rng, mode, it, draws=None):
        w = self.w
        noises = [n for n in (draws[2], draws[3]) if n is not None] if draws is not None else []
        with injected_rng(draws[0] if draws else None, draws[1] if draws else None, noises):
            return w.graph.render(w.opt, self.poses(), H=w.H, W=w.W, intr=w.intr, ray_idx=idx, depth_range=rng, iter=it, mode=mode)

    def step(self, idx, rng, it, draws, target, keep=False):
        w = self.w
        w.optim.zero_grad(set_to_none=True)
        if w.optim_pose is not None:
            w.optim_pose.zero_grad(set_to_none=True)
        ret = self.render(idx, rng, "train", it, draws)
        loss = ops.photometric_loss(ret.rgb, target, rgb_fine=ret.rgb_fine)
        loss.backward()
        if keep:        # gradients AND the weights they belong to, before the optimiser consumes / moves them
            self.kept = (ret, {n: {k: p.grad.detach().clone() for k, p in getattr(w.graph, n).named_parameters() if k != "progress"}
                               for n in ("nerf", "nerf_fine")},
                         {n: {k: v.detach().clone() for k, v in getattr(w.graph, n).state_dict().items()} for n in ("nerf", "nerf_fine")})
        w.optim.step()
        if w.optim_pose is not None:
            w.optim_pose.step()
        return loss.detach()


def photometric_grad_check(tr, idx, rng, it, draws, target, device):
    """parameter-gradient error of trainer `tr` (HipTrainer after step(keep=True), or the fp32 oracle) under the
    photometric loss, against the float64 referee fed the trainer's own rays / depths / draws and weights."""
    B, R = target.shape[:2]
    n_el = B * R * 3
    tgt = target.reshape(1, B * R, 3)

    def loss_fn(part, s):
        t = tgt[:, s].to(part["rgb"].dtype)
        return ((part["rgb"] - t) ** 2).sum() / (n_el + 1e-6) + ((part["rgb_fine"] - t) ** 2).sum() / (n_el + 1e-6)

    if isinstance(tr, HipTrainer):
        ret, got, sds = tr.kept
        sd_c, sd_f = sds["nerf"], sds["nerf_fine"]
        flat = lambda x: x.detach().reshape(1, B * R, *x.shape[2:])
        center, ray, t, t_fine = flat(ret.origins), flat(ret.viewdirs), flat(ret.t), flat(ret.t_fine)
        opt = tr.w.opt
    else:
        return None
    nc = draws[2].reshape(1, B * R, -1) if draws[2] is not None else None
    nf = draws[3].reshape(1, B * R, -1) if draws[3] is not None else None
    _, gref, _, _ = referee(opt, sd_c, sd_f, center, ray, t, t_fine, nc, nf, {"rgb": 1.0}, "train", chunk=1024, want_ray_grad=False,
                            device=device, loss_fn=loss_fn)
    _, g32, _, _ = referee(opt, sd_c, sd_f, center, ray, t, t_fine, nc, nf, {"rgb": 1.0}, "train", chunk=1024, want_ray_grad=False,
                           device=device, loss_fn=loss_fn, dtype=torch.float32)

    def summarise(gg):
        per = {f"{n}.{k}": rel_l2(gg[n][k], gref[n][k]) for n in ("nerf", "nerf_fine") for k in gref[n]}
        a = torch.cat([gg[n][k].detach().double().cpu().reshape(-1) for n in ("nerf", "nerf_fine") for k in gref[n]])
        b = torch.cat([gref[n][k].double().reshape(-1) for n in ("nerf", "nerf_fine") for k in gref[n]])
        worst = max(per, key=per.get)
        return dict(worst_tensor=worst, worst_rel_l2=per[worst], all_params_rel_l2=float((a - b).norm() / b.norm()))

    return dict(hip=summarise(got), reference_fp32=summarise(g32))


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=1, choices=[1, 2, 3])
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--modes", default="bf16x3,bf16,fp32")
    ap.add_argument("--eval-every", type=int, default=250)
    ap.add_argument("--eval-rays", type=int, default=4096, help="held-out rays per view")
    ap.add_argument("--grad-check-at", type=int, default=1000)
    ap.add_argument("--max-seconds", type=float, default=1500.0)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--seed", type=int, default=0, help="shifts the initial weights / initial pose noise and the ray / draw stream of the run")
    ap.add_argument("--out", default=None)
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--device", default="cuda:0", help="cpu: plumbing check of the oracle side only (use --modes '')")
    return ap.parse_args(argv)


def run(args, dev=None):
    """the side-by-side run described in the module docstring; returns the result document"""
    dev = dev or torch.device(args.device)
    say = (lambda *a, **k: None) if args.quiet else print
    torch.backends.cuda.matmul.allow_tf32 = False
    modes = [m for m in args.modes.split(",") if m]
    hips = {m: HipTrainer(args.config, m, dev, args.rays, args.steps, args.seed) for m in modes}
    w0 = next(iter(hips.values())).w if hips else HipTrainer(args.config, "fp32", dev, args.rays, args.steps, args.seed).w
    for m, t in hips.items():      # identical initialisation by construction (same seed); verify
        for a, b in zip(t.w.graph.nerf.parameters(), w0.graph.nerf.parameters()):
            assert torch.equal(a, b)
    oracle = None if args.no_oracle else OracleTrainer(w0, dev)
    trainers = dict(hips)
    if oracle is not None:
        trainers["oracle_fp32"] = oracle
    B, H, W = w0.B, w0.H, w0.W
    R = args.rays // B
    Nc, Nf = w0.opt.nerf.sample_intvs, w0.opt.nerf.sample_intvs_fine
    # (renderer.py:97-108: opt.nerf.depth.range for inverse depth, the data's range otherwise)
    rng = w0.opt.nerf.depth.range if w0.opt.nerf.depth.param == "inverse" else w0.data.depth_range[0]
    use_noise = bool(w0.opt.nerf.density_noise_reg)
    gen = torch.Generator(device=dev).manual_seed(1234 + args.seed)
    held = torch.randperm(H * W, generator=gen, device=dev)[:args.eval_rays]
    held_tgt = w0.img_flat[:, held]
    pose_gt = w0.data.pose

    def evaluate(step):
        row = dict(step=step)
        with torch.no_grad():
            for name, tr in trainers.items():
                outs = []
                for c in range(0, held.numel(), 2048):                      # oracle memory: eval in ray chunks
                    o = tr.render(held[c:c + 2048], rng, "val", None)
                    outs.append(o["rgb_fine"])
                row[name] = dict(psnr=psnr_of(torch.cat(outs, dim=1), held_tgt))
                if args.config != 1:
                    row[name]["pose_err_deg"], row[name]["pose_err_dist"] = pose_error(tr.poses().detach(), pose_gt)
        return row

    curve, grad_checks, losses = [evaluate(0)], {}, {k: [] for k in trainers}
    say(json.dumps(curve[-1]), flush=True)
    t_start = time.perf_counter()
    done = 0
    for it in range(args.steps):
        idx = torch.randperm(H * W, generator=gen, device=dev)[:R]
        jitter = torch.rand(B, R, Nc, 1, generator=gen, device=dev)
        grid = torch.rand(Nf + 1, generator=gen, device=dev)
        nc = torch.randn(B, R, Nc, generator=gen, device=dev) if use_noise else None
        nf = torch.randn(B, R, Nc + Nf, generator=gen, device=dev) if use_noise else None
        draws = (jitter, grid, nc, nf)
        target = w0.img_flat[:, idx]
        check = args.grad_check_at >= 0 and it in (0, args.grad_check_at)
        for name, tr in trainers.items():
            if args.config != 1:
                tr.set_progress(it / args.steps)
            if isinstance(tr, HipTrainer):
                loss = tr.step(idx, rng, it, draws, target, keep=check)
                if check:
                    grad_checks.setdefault(str(it), {})[name] = photometric_grad_check(tr, idx, rng, it, draws, target, dev)
                    tr.kept = None
            else:
                loss = tr.step(idx, rng, it, draws, target)
            if it % 50 == 0 or it == args.steps - 1:
                losses[name].append((it, float(loss)))
        done = it + 1
        if done % args.eval_every == 0 or done == args.steps:
            curve.append(evaluate(done))
            curve[-1]["seconds"] = round(time.perf_counter() - t_start, 1)
            say(json.dumps(curve[-1]), flush=True)
        if time.perf_counter() - t_start > args.max_seconds:
            break
    if curve[-1]["step"] != done:
        curve.append(evaluate(done))
    final = curve[-1]
    ref = "oracle_fp32" if oracle is not None else "fp32"
    if ref not in trainers:
        raise SystemExit("nothing to compare against: keep the oracle or include the fp32 mode")
    delta = {k: final[k]["psnr"] - final[ref]["psnr"] for k in trainers if k != ref}
    # the spread an fp32-level rounding difference alone produces over the run: HIP fp32 vs the fp32 oracle
    doc = dict(what="held-out PSNR (rgb_fine) of HIP precision modes vs the fp32 oracle trained side by side on the GPU from identical "
                    "initialisation with identical rays and draws", config=args.config, steps_done=done, rays_per_step=B * R, samples=f"{Nc}+{Nf}",
               scene=f"analytic sphere scene of bench_workloads.py, {B} views {H}x{W}", reference=ref, final=final, psnr_delta_vs_reference=delta,
               noise_floor="psnr_delta of fp32 (fp32-level arithmetic on a different summation order) is the run-to-run noise of this comparison",
               curve=curve, photometric_grad_error_vs_float64_referee=grad_checks, losses=losses,
               seconds=round(time.perf_counter() - t_start, 1), torch=torch.__version__)
    say(json.dumps(dict(final=final, psnr_delta_vs_reference=delta, grad=grad_checks)), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(doc, f, indent=1)
    return doc


if __name__ == "__main__":
    run(parse())
