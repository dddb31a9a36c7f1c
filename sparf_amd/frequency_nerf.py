"""`source.models.frequency_nerf` of the reference, backed by the MI355X HIP renderer.

Public surface kept from /root/reference/source/models/frequency_nerf.py:
`FrequencyEmbedder(opt)(opt, input, L)` (:42-69) and `NeRF(opt, is_fine_network)` (:72-343)
with `.mlp_feat`, `.mlp_rgb` (ModuleLists of nn.Linear, same state_dict keys), `.progress`,
`initialize()`, `forward`, `forward_samples`, `composite`, `positional_encoding`.
Parameters stay ordinary nn.Parameters in nn.Linear layout, so optimisers, grad clipping,
`load_state_dict(strict=True)` and `progress.data.fill_()` of the reference trainers work
unchanged; the HIP kernels read a packed copy of the weights that is rebuilt when one changes,
and band weights computed from the device value of `progress` on every pass.

The sample -> encode -> MLP -> sigma/rgb -> composite chain runs as one fused pass
(`render_pass`); `forward_samples` + `composite`, which the reference calls back to back
(renderer.py:304-309), are thin views over it.
"""
import math
import os

import numpy as np
import torch

from . import lib as L
from . import ops

COMPOSITE_KEYS = ("rgb", "rgb_var", "depth", "depth_var", "opacity", "weights", "all_cumulated")
MAX_ROWS_PER_CALL = 1 << 23          # upper limit of sample rows per pass launch when activations are saved (a memory bound, not an addressing one)
MAX_ROWS_INFERENCE = 1 << 24         # without gradients only per-row outputs exist (C ABI limit 2^27)
MIN_ROWS_PER_CALL = 1 << 18


SAFE_ROWS = 1 << 21                  # a training pass up to this size (~19 GB of save area + workspace in bf16-plane modes, ~38 GB in fp32) is granted
#                                      on a CACHED free-memory figure (refreshed every 0.5 s), larger ones on a fresh query
_FREE_CACHE = {}
_PER_ROW = {}


def _free_bytes(device, max_age=0.5):
    """free device memory, from the driver at most every `max_age` seconds (a query costs a driver call; a pass must not pay one
    each time, nor run blind on a nearly full device: ADVICE r04)"""
    import time
    key = str(device)
    now = time.monotonic()
    hit = _FREE_CACHE.get(key)
    if hit is None or now - hit[0] > max_age:
        try:
            free = torch.cuda.mem_get_info(device)[0] if torch.cuda.is_available() else 0
        except Exception:
            free = 0
        hit = _FREE_CACHE[key] = (now, free)
    return hit[1]


def max_rows_per_call(prec=None, device=None, need=None, far=None):
    """Sample rows one launch set may take (`need`: the rows the caller wants).  Without gradients: 2^24.  With gradients a pass
    keeps its save area (sparf_save_bytes) and, in the backward, its gradient workspace (sparf_bwd_workspace_bytes) alive --
    about 4.7 + 4.6 KB per row in the bf16-plane modes, 9.2 + 9.1 KB in fp32 (2.6 + 2.5 KB with 8-bit areas), plus, for passes
    with far rows (`far` = (K, far_prec), n_samples known to the caller as need / rays), the far launch's own save scratch -- so the
    cap is what fits in HALF of the device memory that is free (both networks' passes of a render are alive at once), at most 2^23,
    at least 2^18 rows; opt-independent callers that pass no precision get the conservative fp32 figure.  Requests up to SAFE_ROWS
    are checked against a cached free-memory figure, larger ones against a fresh one.  (Round 3 used a constant 2^23: 76-150 GB.)"""
    if not torch.is_grad_enabled():
        return MAX_ROWS_INFERENCE
    p = L.PREC_FP32 if prec is None else prec
    key = (p, far[1] if (far is not None and isinstance(far[0], int)) else None)
    per_row = _PER_ROW.get(key)
    if per_row is None:
        lib = L.load()
        per_row = (lib.sparf_save_bytes(p, 1 << 16) + lib.sparf_bwd_workspace_bytes(p, 1 << 10, 1 << 6, 1)) / float(1 << 16)
        if key[1] is not None:
            # (K of every n samples once more in far_prec's save format: K <= 8 of >= 64, bounded here by an eighth of a pass in that format)
            per_row += lib.sparf_save_bytes(key[1], 1 << 16) / float(1 << 16) * 0.125
        _PER_ROW[key] = per_row
    small = need is not None and need <= SAFE_ROWS
    free = _free_bytes(device, max_age=0.5 if small else 0.0)
    if small and (not free or 0.5 * free >= need * per_row):
        return SAFE_ROWS
    rows = int(0.5 * free / per_row) if free else MAX_ROWS_PER_CALL
    rows = max(MIN_ROWS_PER_CALL, min(MAX_ROWS_PER_CALL, rows))
    return max(MIN_ROWS_PER_CALL, rows // 8192 * 8192)


DEFAULT_PRECISION = "bf16x3"         # the ONE default: what an unmodified run_trainval.py gets, and what bench.py measures
DEFAULT_FAR_SAMPLES = 8


def _hip_get(opt):
    hip = opt.get("hip", None) if hasattr(opt, "get") else getattr(opt, "hip", None)
    if hip is None:
        return lambda k, d=None: d
    return lambda k, d=None: (hip.get(k, d) if hasattr(hip, "get") else getattr(hip, k, d))


def precision_name(opt):
    """opt.hip.precision, else $SPARF_PRECISION, else DEFAULT_PRECISION ('bf16x3': the fastest mode whose outputs stay
    within 1e-4 of the reference; 'fp32' = exact fp32 MFMA arithmetic, the reference's own floor; 'bf16' = throughput mode)."""
    name = _hip_get(opt)("precision") or os.environ.get("SPARF_PRECISION") or DEFAULT_PRECISION
    if name not in L.PREC_IDS:
        raise ValueError(f"unknown precision {name!r} (choose from {sorted(L.PREC_IDS)})")
    return name


DEFAULT_FAR_DEPTH = 8.0


def pass_precision(opt, n_coarse=None, to_max_samples=None):
    """-> (prec, far): MFMA operand precision of a pass and its far routing -- (K, far_prec): the last K samples of every ray,
    or (threshold, far_prec): the tiles whose depth samples exceed a threshold -- or None.

    bf16x3 promises outputs within 1e-4 of the reference.  It keeps that promise for metric depth; with INVERSE depth
    (`opt.nerf.depth.param == 'inverse'`, renderer.py:413-416) sample i of a ray sits at t = 1 / (1 - (u + i) / N + 1e-8): the
    last one at t = N / (1 - u), up to 1e8, the ones before it at N/2, N/3, ...  The network is evaluated at |p| ~ t, its
    pre-activations grow with it, and a 16-bit-mantissa operand (head + tail) is 128x further from the fp32 result than fp32
    is from float64: on rays that have not terminated before those samples the RENDERED outputs miss 1e-4 (six seeds at
    BASELINE config 3, profiles/r04_inverse_routing_study.json: 4.7e-5 ... 1.1e-4 with every row in bf16x3; 6.8e-5 with the
    last sample in fp32; 3.7e-5 with the last 4; 1.2e-5 with the last 8 -- the level of the metric-depth configs).
    Such passes therefore ROUTE THE LAST K SAMPLES OF EVERY RAY through the fp32 kernels (C ABI "far rows": a second,
    nrays x K-row launch whose raw density / colour replace the bf16x3 ones before compositing; the backward carries those rows'
    gradient through the fp32 dgrad / wgrad) and everything else through bf16x3.  K = opt.hip.far_samples (8), at most the
    coarse sample count - 1.  `n_coarse`: the number of stratified inverse-depth samples at the END of each ray of this pass
    (Graph.render: the coarse samples, also after the merge with the fine ones, which all lie below them -- renderer.py:446
    draws them in the un-inverted range); None for passes whose samples do not have that structure.
    `to_max_samples`: the sample count of a render_to_max pass (renderer.py:595-624: n samples up to a PER-RAY far bound, metric
    spacing from depth.range[0]): which of its samples are far is a property of the data, so such a pass -- always rendered under
    no_grad by its one caller, depth_cons_loss.py:266 -- is routed BY VALUE: every 128-row tile whose largest depth sample exceeds
    opt.hip.far_depth (8: where the stratified samples' K = 8 draws the line) is evaluated by the fp32 inference kernel, every
    other tile by the bf16x3 one (C ABI far_count = -1); needs 32 | n and no gradients.
    Anything else (explicit points, the public forward_samples, render_to_max with gradients) runs on the fp32 kernels as a whole,
    as all inverse-depth passes did in round 3.
    opt.hip.inverse_depth_precision: 'routed' (default) | 'fp32' (whole passes, round 3) | 'bf16x3' (no correction, ~1e-4)."""
    get = _hip_get(opt)
    name = precision_name(opt)
    # '+q8' (8-bit save and gradient areas, lib.SAVE_Q8) changes what a training pass keeps for its backward, not its arithmetic: the
    # routing below is that of the plain mode, and a pass with far ROWS keeps plane saves (the far rows' activations are transplanted
    # into bf16 planes, sparf_hip.h)
    q8 = L.SAVE_Q8 if name.endswith("+q8") else 0
    name = name[:-3] if q8 else name
    if name == "bf16x3" and opt.nerf.depth.param == "inverse":
        how = get("inverse_depth_precision") or os.environ.get("SPARF_INVERSE_DEPTH_PRECISION") or "routed"
        if how not in ("routed", "fp32", "bf16x3"):
            raise ValueError(f"opt.hip.inverse_depth_precision must be 'routed', 'fp32' or 'bf16x3', not {how!r}")
        if how == "bf16x3":
            return L.PREC_X3 | q8, None
        K = int(get("far_samples") or os.environ.get("SPARF_FAR_SAMPLES") or DEFAULT_FAR_SAMPLES)
        if how == "routed" and n_coarse is not None and min(K, n_coarse - 1) > 0:
            return L.PREC_X3, (min(K, n_coarse - 1), L.PREC_FP32)
        if how == "routed" and to_max_samples is not None and to_max_samples % 32 == 0 and not torch.is_grad_enabled():
            return L.PREC_X3, (float(get("far_depth") or os.environ.get("SPARF_FAR_DEPTH") or DEFAULT_FAR_DEPTH), L.PREC_FP32)
        return L.PREC_FP32, None
    return L.PREC_IDS[name] | q8, None


def get_precision(opt):
    """Precision id of a pass without far-row structure (pass_precision(opt)[0])."""
    return pass_precision(opt)[0]


class FrequencyEmbedder:
    """sin/cos positional embedding, frequency_nerf.py:42-69 (elementwise torch; the fused
    kernel has its own in-register version, this one serves callers of the public API)."""

    def __init__(self, opt):
        self.opt = opt

    def __call__(self, opt, input, L):
        pe = opt.arch.posenc
        if pe.log_sampling:
            freq = 2 ** torch.arange(L, dtype=torch.float32, device=input.device)
            if pe.include_pi_in_posenc:
                freq = freq * np.pi
        else:
            freq = torch.linspace(2.0 ** 0.0, 2.0 ** (L - 1), steps=L, device=input.device) * np.pi
        spectrum = input[..., None] * freq
        enc = torch.stack([spectrum.sin(), spectrum.cos()], dim=-2)
        return enc.view(*input.shape[:-1], -1)


class NeRF(torch.nn.Module):
    def __init__(self, opt, is_fine_network=False):
        super().__init__()
        self.opt = opt
        self.is_fine_network = is_fine_network
        self.define_network(opt, is_fine_network=is_fine_network)
        # Parameter so that it is checkpointed (frequency_nerf.py:79-85)
        self.progress = torch.nn.Parameter(torch.tensor(1.0 if opt.barf_c2f is None else 0.0))
        self._packed = {}

    # ------------------------------------------------------------------ construction
    def define_network(self, opt, is_fine_network=False):
        pe = opt.arch.posenc
        d3 = (3 if pe.add_raw_3D_points else 0) + (6 * pe.L_3D if pe.L_3D > 0 else 0)
        dv = ((3 if pe.add_raw_rays else 0) + (6 * pe.L_view if pe.L_view > 0 else 0)) if opt.nerf.view_dep else 0
        layers_feat = opt.arch.layers_feat
        if is_fine_network and opt.arch.get("layers_feat_fine", None) is not None:
            layers_feat = opt.arch.layers_feat_fine
        self.mlp_feat = torch.nn.ModuleList()
        n = len(layers_feat) - 1
        for li in range(n):
            k_in = d3 if li == 0 else layers_feat[li]
            if li in opt.arch.skip:
                k_in += d3
            k_out = layers_feat[li + 1] + (1 if li == n - 1 else 0)
            lin = torch.nn.Linear(k_in, k_out)
            if opt.arch.tf_init:
                self.tensorflow_init_weights(opt, lin, out="first" if li == n - 1 else None)
            self.mlp_feat.append(lin)
        self.mlp_rgb = torch.nn.ModuleList()
        lr = opt.arch.layers_rgb
        for li in range(len(lr) - 1):
            k_in = layers_feat[-1] + dv if li == 0 else lr[li]
            lin = torch.nn.Linear(k_in, lr[li + 1])
            if opt.arch.tf_init:
                self.tensorflow_init_weights(opt, lin, out="all" if li == len(lr) - 2 else None)
            self.mlp_rgb.append(lin)
        shapes = [tuple(m.weight.shape) for m in list(self.mlp_feat) + list(self.mlp_rgb)]
        if shapes != L.LAYER_SHAPES or opt.arch.density_activ != "softplus" or not pe.log_sampling \
                or not pe.include_pi_in_posenc or list(opt.arch.skip) != [4]:
            raise NotImplementedError(
                "sparf_amd compiles the shipped SPARF architecture only (8x256 feature MLP, skip [4], L_3D=10, "
                f"L_view=4, raw inputs, softplus density, 128-wide colour branch); got layer shapes {shapes}")

    def initialize(self):
        for m in self.modules():
            if isinstance(m, torch.nn.Linear):
                self.tensorflow_init_weights(self.opt, m)

    def tensorflow_init_weights(self, opt, linear, out=None):
        """Xavier-uniform with relu gain except the sigma row / rgb output (frequency_nerf.py:136-147)."""
        gain = torch.nn.init.calculate_gain("relu")
        if out == "all":
            torch.nn.init.xavier_uniform_(linear.weight)
        elif out == "first":
            torch.nn.init.xavier_uniform_(linear.weight[:1])
            torch.nn.init.xavier_uniform_(linear.weight[1:], gain=gain)
        else:
            torch.nn.init.xavier_uniform_(linear.weight, gain=gain)
        torch.nn.init.zeros_(linear.bias)

    # ------------------------------------------------------------------ HIP plumbing
    def hip_params(self):
        """the 20 parameter tensors W0, b0, ..., W9, b9 (read from the modules' parameter dicts: `m.weight` goes through
        nn.Module.__getattr__, 40 of those per render call were ~40 us of host time)"""
        out = []
        for ml in (self.mlp_feat, self.mlp_rgb):
            for m in ml:
                ps = m._parameters
                out.append(ps["weight"])
                out.append(ps["bias"])
        return out

    def flat_params(self, params=None):
        """All 20 parameter tensors as ONE flat autograd tensor (ops.FlatParams: one `cat`), made once per weight version
        and shared by every render call until the weights change.  ops.RenderFn takes it as its parameter input and returns ONE flat
        gradient per network: autograd then sums the gradients of an iteration's render calls with one add per network and call
        instead of one per parameter tensor and call (the unmodified SPARF losses issue six render calls per iteration: 200 tiny
        add launches per iteration with 40 parameter inputs per call), and the cat's backward hands each parameter its view of that
        sum -- `p.grad` stays what it was: views tiling one flat buffer per network (optim.FusedAdam, parallel.GradBucket), and
        `torch.autograd.grad(loss, params)` works as before.  The value of the tensor is not read by the kernels (they read the
        packed weight streams): it is the route of the gradient."""
        params = self.hip_params() if params is None else params
        if not any(p.requires_grad for p in params):
            return None
        key = tuple((p.data_ptr(), p._version, p.requires_grad) for p in params) + (getattr(self, "_weights_epoch", 0),)
        hit = getattr(self, "_flat", None)
        if hit is None or hit[0] != key:
            hit = self._flat = (key, ops.FlatParams.apply(*params))
        return hit[1]

    def release_autograd_cache(self):
        """Forget the cached flat parameter tensor (flat_params).  It keeps the autograd graph of the last iteration's parameter
        route -- and with it the parameters' AccumulateGrad nodes, which remember the stream they were created on -- alive until the
        weights change; call this when the stream changes under a model (before capturing a step in a hipGraph on another stream)."""
        self._flat = None

    def weights_changed(self):
        """Tell the packed-weight cache that WEIGHT values were modified by something torch's
        version counters do not see: a raw-pointer kernel such as optim.FusedAdam, or a write
        through `.data` (`p.data.copy_()` leaves `p._version` alone).  `progress` needs no such
        call: the band weights are recomputed from its device value on every pass."""
        self._weights_epoch = getattr(self, "_weights_epoch", 0) + 1
        self._flat = None                  # (its key holds the epoch: the entry could never hit again, only pin the last iteration's graph)

    def train(self, mode=True):
        """nn.Module.train / eval, plus: the cached flat parameter proxy of the previous phase is dropped (it is a non-leaf
        autograd tensor that kept the last training iteration's parameter route alive through a whole validation run)"""
        self._flat = None
        return super().train(mode)

    def __getstate__(self):
        """copy.deepcopy / pickle: everything but the cached flat parameter proxy -- a NON-LEAF autograd tensor, which torch refuses to
        deep-copy ("Only Tensors created explicitly by the user support the deepcopy protocol") -- and the packed weight streams,
        which are rebuilt from the copied parameters on first use"""
        state = dict(self.__dict__)
        state["_flat"] = None
        state["_packed"] = {}
        return state

    def packed(self, prec, params=None):
        """Packed MFMA weight streams for the current weight values, cached on the tensors'
        version counters (optimiser steps, `load_state_dict`, `re_initialize` all bump them):
        one pack per optimiser step.  The BARF band weights are NOT in here (band_weights()).
        params: hip_params() if the caller has them already (20 nn.Module attribute lookups)."""
        prec = L.base_prec(prec)
        params = self.hip_params() if params is None else params
        key = tuple((p.data_ptr(), p._version) for p in params) + (getattr(self, "_weights_epoch", 0),)
        hit = self._packed.get(prec)
        if hit is None or hit[0] != key:
            blob = ops.pack_weights(params, prec)   # fresh blob: an older one may still be saved for a pending backward
            hit = (key, blob)
            self._packed[prec] = hit
        return hit[1]

    def band_weights(self):
        """Coarse-to-fine band weights for the pass about to run, from the device value of
        `progress` right now (frequency_nerf.py:248-253 reads `self.progress.data` per call)."""
        return ops.c2f_weights(self.progress, self.opt.barf_c2f, self.progress.device)

    def render_pass(self, opt, center, ray, depth_samples, mode=None, noise=None, n_coarse=None, to_max=False):
        """center, ray [B,R,3]; depth_samples [B,R,N,1] (or [B,R,N]).  Returns the union of
        the reference's `forward_samples` and `composite` dictionaries, reference shapes.
        n_coarse: the stratified coarse samples at the end of every ray (Graph.render passes it); to_max: a render_to_max pass
        (pass_precision)."""
        B, R = ray.shape[:2]
        N = depth_samples.shape[2]
        t = depth_samples.reshape(B * R, N)
        prec, far = pass_precision(opt, n_coarse, to_max_samples=N if to_max else None)
        params = self.hip_params()
        far = (far[0], far[1], self.packed(far[1], params)) if far is not None else None
        use_noise = bool(opt.nerf.density_noise_reg) and mode == "train"
        if use_noise and noise is None:
            noise = torch.randn(B * R, N, device=ray.device)       # frequency_nerf.py:192
        c, d = center.reshape(B * R, 3), ray.reshape(B * R, 3)
        nz = noise.reshape(B * R, N) if use_noise else None
        args = (float(opt.nerf.density_noise_reg) if use_noise else 0.0, bool(opt.nerf.setbg_opaque or opt.mask_img),
                prec, self.packed(prec, params), self.band_weights(), params)
        max_rays = max(1, max_rows_per_call(prec, ray.device, need=B * R * N, far=far) // N)
        if B * R <= max_rays:
            out = ops.nerf_pass(c, d, t, nz, *args, far=far)
        else:
            # more sample rows than one launch set is given memory for (max_rows_per_call): consecutive ray chunks
            # (rays are independent; autograd sums the parameter gradients), outputs concatenated
            parts = [ops.nerf_pass(c[i:i + max_rays], d[i:i + max_rays], t[i:i + max_rays],
                                   nz[i:i + max_rays] if nz is not None else None, *args, far=far) for i in range(0, B * R, max_rays)]
            out = {k: torch.cat([p[k] for p in parts], dim=0) for k in parts[0]}
        return dict(rgb_samples=out["rgb_samples"].view(B, R, N, 3), density_samples=out["density_samples"].view(B, R, N),
                    rgb=out["rgb"].view(B, R, 3), rgb_var=out["rgb_var"].view(B, R, 1), depth=out["depth"].view(B, R, 1),
                    depth_var=out["depth_var"].view(B, R, 1), opacity=out["opacity"].view(B, R, 1),
                    weights=out["weights"].view(B, R, N, 1), all_cumulated=out["all_cumulated"].view(B, R))

    # ------------------------------------------------------------------ reference API
    def forward_samples(self, opt, center, ray, depth_samples, embedder_pts, embedder_view, mode=None):
        """frequency_nerf.py:260-281.  Runs the fused pass; the compositing results ride
        along under a private key and are surfaced by `composite`."""
        full = self.render_pass(opt, center, ray, depth_samples, mode=mode)
        pred = dict(rgb_samples=full["rgb_samples"], density_samples=full["density_samples"])
        pred["_fused"] = (depth_samples, {k: full[k] for k in COMPOSITE_KEYS}, pred["rgb_samples"], pred["density_samples"])
        return pred

    def composite(self, opt, ray, pred_dict, depth_samples):
        """frequency_nerf.py:283-343.  A dictionary that `forward_samples` produced for the same depth samples (the way the
        reference calls it, renderer.py:304-309) already carries the fused pass's compositing results; any other dictionary
        with `rgb_samples` [B,R,N,3] and `density_samples` [B,R,N] -- the function is a free function of its arguments in the
        reference -- is composited by the stand-alone kernels (ops.Composite, C ABI 6): every output differentiable w.r.t. the
        per-sample values and the ray."""
        fused = pred_dict.pop("_fused", None)
        if fused is not None and fused[0] is depth_samples and fused[2] is pred_dict.get("rgb_samples") and fused[3] is pred_dict.get("density_samples"):
            pred_dict.update(fused[1])
            return pred_dict
        rgb_s, dens = pred_dict["rgb_samples"], pred_dict["density_samples"]
        B, R, N = dens.shape
        out = ops.composite(ray.reshape(B * R, 3), dens.reshape(B * R, N), rgb_s.reshape(B * R, N, 3), depth_samples.reshape(B * R, N),
                            bool(opt.nerf.setbg_opaque or opt.mask_img))
        pred_dict.update(rgb=out["rgb"].view(B, R, 3), rgb_var=out["rgb_var"].view(B, R, 1), depth=out["depth"].view(B, R, 1),
                         depth_var=out["depth_var"].view(B, R, 1), opacity=out["opacity"].view(B, R, 1), weights=out["weights"].view(B, R, N, 1),
                         all_cumulated=out["all_cumulated"].view(B, R))
        return pred_dict

    def forward(self, opt, points_3D_samples, ray, embedder_pts, embedder_view, mode=None):
        """frequency_nerf.py:172-226 for explicitly given points [B,R,N,3]: every point is
        evaluated as a one-sample ray starting at the point (p = c + r*0)."""
        B, R, N = points_3D_samples.shape[:3]
        c = points_3D_samples.reshape(1, B * R * N, 3)
        r = ray[:, :, None, :].expand(B, R, N, 3).reshape(1, B * R * N, 3)
        t = torch.zeros(1, B * R * N, 1, 1, device=ray.device)
        full = self.render_pass(opt, c, r, t, mode=mode)
        return dict(rgb_samples=full["rgb_samples"].view(B, R, N, 3), density_samples=full["density_samples"].view(B, R, N))

    def positional_encoding(self, opt, input, embedder_fn, L):
        """BARF coarse-to-fine masked encoding, frequency_nerf.py:229-258 (public helper)."""
        enc = embedder_fn(opt, input, L)
        if opt.barf_c2f is not None:
            start, end = opt.barf_c2f
            alpha = (self.progress.data - start) / (end - start) * L
            k = torch.arange(L, dtype=torch.float32, device=input.device)
            weight = (1 - (alpha - k).clamp_(min=0, max=1).mul_(math.pi).cos_()) / 2
            shape = enc.shape
            enc = (enc.view(-1, L) * weight).view(*shape)
        return enc
