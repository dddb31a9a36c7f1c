"""What would 8-bit saved activations / gradients cost?  ("next" item 1: the bytes.)

    python tests/tools/save_precision_study.py --steps 1200 --out gpurun_out/r03_save_precision_study.json

The step is HBM-structured: the forward saves 3.8 GB of layer inputs per 786 k-row pass, dgrad writes 3.5 GB of layer-output
gradients, wgrad reads both back (7.2 GB, 5.3 TB/s = its whole run time).  Halving those bytes needs an fp8 wgrad and fp8
savers -- days of kernel work -- so this script first answers whether the ARITHMETIC would be acceptable, by emulation:
the fp32 oracle (oracle/nerf_oracle.py as PyTorch-ROCm ops) with its `F.linear` replaced by an autograd function that
rounds exactly the operands the HIP kernels round, trained side by side with the plain fp32 oracle from identical
initialisation on identical rays and draws (the protocol of tests/tools/psnr_curve.py):

  emu_bf16x3          forward fp32 (the head + tail forward is 1e-5 from it), propagated gradient rounded to bf16 before it
                      multiplies the weights (dgrad), wgrad operands X and dY rounded to bf16      = today's headline mode
  emu_bf16x3_x8       ... saved X in e4m3 with a power-of-two scale per 32-row tile
  emu_bf16x3_x8_dy8   ... and the saved dY in e5m2, same scaling
  emu_bf16x3_xi8t / _xi8r / _xi8r_dyi8r   the same with a LINEAR 8-bit grid (one scale per 32-row tile / per row)
  emu_bf16            operands of all three products in bf16                                        = today's bf16 mode
  emu_bf16_x8_dy8     ... with the 8-bit saves

Reported: the parameter-gradient error of every variant against the fp32 oracle at step 0 (same weights, photometric loss),
and the held-out PSNR curves.  Test infrastructure (imports oracle/); nothing here is a product path.
"""
import argparse
import json
import os
import sys
import time
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]

from oracle import nerf_oracle as O                                        # noqa: E402
from tests.tools.psnr_curve import HipTrainer, OracleTrainer, psnr_of    # noqa: E402


def q_bf16(x):
    return x.bfloat16().float()


def q_fp8_tiles(x, mant, emin, vmax, tile=32):
    """round a [rows, K] matrix to an 8-bit float (mant mantissa bits, smallest normal exponent emin, largest value vmax)
    after scaling every `tile`-row block by a power of two that brings its largest magnitude just under vmax"""
    rows, K = x.shape
    pad = (-rows) % tile
    xp = torch.nn.functional.pad(x, (0, 0, 0, pad)) if pad else x
    t = xp.reshape(-1, tile * K)
    amax = t.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)
    scale = torch.exp2(torch.ceil(torch.log2(amax / vmax)))
    v = t / scale
    a = v.abs().clamp(max=vmax)
    e = torch.floor(torch.log2(a.clamp_min(2.0 ** (emin - mant)))).clamp_min(emin)
    step = torch.exp2(e - mant)
    q = torch.round(a / step) * step
    out = (torch.sign(v) * q * scale).reshape(-1, K)
    return out[:rows] if pad else out


def q_int8(x, rows_per_scale):
    """symmetric linear 8-bit grid (step = amax / 127) with one scale per `rows_per_scale` rows of a [rows, K] matrix: an absolute
    error that is uniform over the block -- what a SUM over rows (the weight gradient) cares about -- instead of a relative one"""
    rows, K = x.shape
    pad = (-rows) % rows_per_scale
    xp = torch.nn.functional.pad(x, (0, 0, 0, pad)) if pad else x
    t = xp.reshape(-1, rows_per_scale * K)
    step = t.abs().amax(dim=1, keepdim=True).clamp_min(1e-30) / 127.0
    out = (torch.round(t / step) * step).reshape(-1, K)
    return out[:rows] if pad else out


QUANT = {
    "fp32": lambda x: x,
    "bf16": q_bf16,
    "e4m3": lambda x: q_fp8_tiles(x, 3, -6, 448.0),
    "e5m2": lambda x: q_fp8_tiles(x, 2, -14, 57344.0),
    "i8t": lambda x: q_int8(x, 32),          # one scale per 32-row tile
    "i8r": lambda x: q_int8(x, 1),           # one scale per row
}


class EmuLinear(torch.autograd.Function):
    """y = q_fwd(x) q_fwd(W)^T + b;  dX = q_dy(dY) q_dw(W);  dW = q_sdy(dY)^T q_sx(x);  db = column sums of q_sdy(dY)"""

    @staticmethod
    def forward(ctx, x, w, b, cfg):
        xf = x.reshape(-1, x.shape[-1])
        y = QUANT[cfg.fwd](xf) @ QUANT[cfg.fwd](w).t() + b
        ctx.cfg, ctx.shape = cfg, x.shape
        ctx.save_for_backward(QUANT[cfg.save_x](xf), w)
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        xs, w = ctx.saved_tensors
        cfg = ctx.cfg
        d = dy.reshape(-1, dy.shape[-1])
        dx = QUANT[cfg.dy](d) @ QUANT[cfg.dgrad_w](w)
        ds = QUANT[cfg.save_dy](d)
        return dx.reshape(ctx.shape), ds.t() @ xs, ds.sum(0), None


class _ShimF:
    """stands in for torch.nn.functional inside oracle.nerf_oracle while an emulated trainer renders"""

    def __init__(self, cfg):
        self.cfg = cfg

    def linear(self, x, w, b):
        return EmuLinear.apply(x, w, b, self.cfg)

    def __getattr__(self, k):
        return getattr(torch.nn.functional, k)


VARIANTS = {
    "emu_bf16x3": dict(fwd="fp32", dy="bf16", dgrad_w="fp32", save_x="bf16", save_dy="bf16"),
    "emu_bf16x3_x8": dict(fwd="fp32", dy="bf16", dgrad_w="fp32", save_x="e4m3", save_dy="bf16"),
    "emu_bf16x3_x8_dy8": dict(fwd="fp32", dy="bf16", dgrad_w="fp32", save_x="e4m3", save_dy="e5m2"),
    "emu_bf16x3_xi8t": dict(fwd="fp32", dy="bf16", dgrad_w="fp32", save_x="i8t", save_dy="bf16"),
    "emu_bf16x3_xi8r": dict(fwd="fp32", dy="bf16", dgrad_w="fp32", save_x="i8r", save_dy="bf16"),
    "emu_bf16x3_xi8r_dyi8r": dict(fwd="fp32", dy="bf16", dgrad_w="fp32", save_x="i8r", save_dy="i8r"),
    "emu_bf16x3_xi8t_dyi8t": dict(fwd="fp32", dy="bf16", dgrad_w="fp32", save_x="i8t", save_dy="i8t"),
    "emu_bf16": dict(fwd="bf16", dy="bf16", dgrad_w="bf16", save_x="bf16", save_dy="bf16"),
    "emu_bf16_xi8t_dyi8t": dict(fwd="bf16", dy="bf16", dgrad_w="bf16", save_x="i8t", save_dy="i8t"),
    "emu_bf16_x8_dy8": dict(fwd="bf16", dy="bf16", dgrad_w="bf16", save_x="e4m3", save_dy="e5m2"),
}


class EmuTrainer(OracleTrainer):
    def __init__(self, w, device, cfg):
        super().__init__(w, device)
        self.shim = _ShimF(types.SimpleNamespace(**cfg))

    def render(self, idx, rng, mode, it, draws=None):
        keep = O.F
        O.F = self.shim
        try:
            return super().render(idx, rng, mode, it, draws)
        finally:
            O.F = keep


def grads_of(tr):
    return {f"{n}.{k}": v.grad.detach().clone() for n, p in (("nerf", tr.pc), ("nerf_fine", tr.pf)) for k, v in p.items() if v.grad is not None}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1200)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--eval-every", type=int, default=300)
    ap.add_argument("--eval-rays", type=int, default=4096)
    ap.add_argument("--variants", default=",".join(VARIANTS))
    ap.add_argument("--max-seconds", type=float, default=1200.0)
    ap.add_argument("--out", default=None)
    args = ap.parse_args(argv)
    dev = torch.device("cuda:0")
    torch.backends.cuda.matmul.allow_tf32 = False
    w0 = HipTrainer(1, "fp32", dev, args.rays, args.steps).w            # the scene / initialisation of psnr_curve config 1
    trainers = {"oracle_fp32": OracleTrainer(w0, dev)}
    for name in [v for v in args.variants.split(",") if v]:
        trainers[name] = EmuTrainer(w0, dev, VARIANTS[name])
    B, H, W = w0.B, w0.H, w0.W
    R = args.rays // B
    Nc, Nf = w0.opt.nerf.sample_intvs, w0.opt.nerf.sample_intvs_fine
    rng = w0.data.depth_range[0]
    use_noise = bool(w0.opt.nerf.density_noise_reg)
    gen = torch.Generator(device=dev).manual_seed(1234)
    held = torch.randperm(H * W, generator=gen, device=dev)[:args.eval_rays]
    held_tgt = w0.img_flat[:, held]

    def evaluate(step):
        row = dict(step=step)
        with torch.no_grad():
            for name, tr in trainers.items():
                outs = [tr.render(held[c:c + 2048], rng, "val", None)["rgb_fine"] for c in range(0, held.numel(), 2048)]
                row[name] = psnr_of(torch.cat(outs, dim=1), held_tgt)
        return row

    def direction_errors(grad_sets):
        ref_g, out = grad_sets["oracle_fp32"], {}
        for name, g in grad_sets.items():
            if name == "oracle_fp32":
                continue
            per = {k: float((g[k].double() / g[k].double().norm() - ref_g[k].double() / ref_g[k].double().norm()).norm()) for k in ref_g}
            a = torch.cat([g[k].double().reshape(-1) for k in ref_g])
            b = torch.cat([ref_g[k].double().reshape(-1) for k in ref_g])
            worst = max(per, key=per.get)
            out[name] = dict(worst_tensor=worst, worst_direction_err=per[worst], all_params_direction_err=float((a / a.norm() - b / b.norm()).norm()))
        return out

    # the same gradients under the parity tests' kind of loss: a random linear functional of the rendered colours, whose per-row
    # terms do not add coherently (tests/scale_cases.py) -- initial weights, one backward per trainer, no update
    idx0 = torch.randperm(H * W, generator=gen, device=dev)[:R]
    draws0 = (torch.rand(B, R, Nc, 1, generator=gen, device=dev), torch.rand(Nf + 1, generator=gen, device=dev),
              torch.randn(B, R, Nc, generator=gen, device=dev) if use_noise else None,
              torch.randn(B, R, Nc + Nf, generator=gen, device=dev) if use_noise else None)
    Gc, Gf = torch.rand(B, R, 3, generator=gen, device=dev) * 2 - 1, torch.rand(B, R, 3, generator=gen, device=dev) * 2 - 1
    sets = {}
    for name, tr in trainers.items():
        tr.optim.zero_grad(set_to_none=True)
        out = tr.render(idx0, rng, "train", 0, draws0)
        ((out["rgb"] * Gc).sum() + (out["rgb_fine"] * Gf).sum()).backward()
        sets[name] = grads_of(tr)
        tr.optim.zero_grad(set_to_none=True)
    grad_err_random = direction_errors(sets)
    print(json.dumps(dict(grad_err_random_linear_loss=grad_err_random)), flush=True)
    if args.steps == 0:                       # gradient errors only: the photometric loss on the same batch, no update either
        target0 = w0.img_flat[:, idx0]
        for name, tr in trainers.items():
            out = tr.render(idx0, rng, "train", 0, draws0)
            e1, e2 = (out["rgb"] - target0) ** 2, (out["rgb_fine"] - target0) ** 2
            (e1.sum() / (e1.nelement() + 1e-6) + e2.sum() / (e2.nelement() + 1e-6)).backward()
            sets[name] = grads_of(tr)
            tr.optim.zero_grad(set_to_none=True)
        grad_err_photo = direction_errors(sets)
        print(json.dumps(dict(grad_err_photometric_loss=grad_err_photo)), flush=True)
        if args.out:
            with open(args.out, "w") as f:
                json.dump(dict(variants={k: VARIANTS[k] for k in trainers if k in VARIANTS},
                               grad_direction_error_random_linear_loss_vs_fp32_oracle=grad_err_random,
                               grad_direction_error_photometric_loss_vs_fp32_oracle=grad_err_photo), f, indent=1)
        return

    curve, grad_err = [evaluate(0)], {}
    print(json.dumps(curve[-1]), flush=True)
    t0 = time.perf_counter()
    done = 0
    for it in range(args.steps):
        idx = torch.randperm(H * W, generator=gen, device=dev)[:R]
        draws = (torch.rand(B, R, Nc, 1, generator=gen, device=dev), torch.rand(Nf + 1, generator=gen, device=dev),
                 torch.randn(B, R, Nc, generator=gen, device=dev) if use_noise else None,
                 torch.randn(B, R, Nc + Nf, generator=gen, device=dev) if use_noise else None)
        target = w0.img_flat[:, idx]
        ref_g = None
        for name, tr in trainers.items():
            tr.step(idx, rng, it, draws, target)
            if it == 0:                                                  # gradients of step 0: identical weights in every trainer
                g = grads_of(tr)                                         # (clip_grad_norm_ scaled them: compare directions and norms apart)
                if name == "oracle_fp32":
                    ref_g = g
                else:
                    per = {}
                    for k in ref_g:
                        a, b = g[k].double(), ref_g[k].double()
                        per[k] = float((a / a.norm() - b / b.norm()).norm())            # clip-invariant: error of the direction
                    a = torch.cat([g[k].double().reshape(-1) for k in ref_g])
                    b = torch.cat([ref_g[k].double().reshape(-1) for k in ref_g])
                    worst = max(per, key=per.get)
                    grad_err[name] = dict(worst_tensor=worst, worst_direction_err=per[worst], all_params_direction_err=float((a / a.norm() - b / b.norm()).norm()))
        if it == 0:
            print(json.dumps(dict(grad_err_step0=grad_err)), flush=True)
        done = it + 1
        if done % args.eval_every == 0 or done == args.steps:
            curve.append(evaluate(done))
            curve[-1]["seconds"] = round(time.perf_counter() - t0, 1)
            print(json.dumps(curve[-1]), flush=True)
        if time.perf_counter() - t0 > args.max_seconds:
            break
    final = curve[-1]
    doc = dict(what="emulated operand roundings of the HIP modes, with and without 8-bit saved activations / gradients, trained side by side "
                    "with the fp32 oracle (psnr_curve.py protocol, config 1)", variants={k: VARIANTS[k] for k in trainers if k in VARIANTS},
               steps_done=done, rays_per_step=B * R, final=final, psnr_delta_vs_oracle={k: final[k] - final["oracle_fp32"] for k in trainers if k != "oracle_fp32"},
               grad_direction_error_step0_vs_fp32_oracle=grad_err, grad_direction_error_random_linear_loss_vs_fp32_oracle=grad_err_random, curve=curve, seconds=round(time.perf_counter() - t0, 1))
    print(json.dumps(dict(final=final, delta=doc["psnr_delta_vs_oracle"])), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
