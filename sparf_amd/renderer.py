"""`source.models.renderer.Graph` of the reference, on the MI355X HIP hot path.

Same class name, constructor, attributes, method names and signatures as
/root/reference/source/models/renderer.py:28-624, same `EasyDict` results (keys, shapes,
`_fine` suffixes), same `opt.*` keys.  What differs is underneath:
rays are generated only for the requested pixels, depth samples / resampling+sort / the
two MLP passes / compositing are HIP kernels behind libsparf_hip.so, and full images are
rendered in large chunks instead of `rand_rays`-sized slices.
"""
import logging

import numpy as np
import torch

from . import camera
from . import lib as L
from . import ops
from .edict import EasyDict as edict
from .frequency_nerf import FrequencyEmbedder, NeRF, max_rows_per_call, pass_precision


_LOGGED_MODES = set()


def _as_float(x):
    return float(x.item()) if torch.is_tensor(x) else float(x)


class _Shifted:
    """rows [lo, hi) of a group addressed in a buffer that starts at group row `lo` (render_batch)"""

    def __init__(self, buf, lo):
        self.buf, self.lo = buf, lo

    def __getitem__(self, s):
        return self.buf[s.start - self.lo:s.stop - self.lo]


class PendingRender(edict):
    """The result of a render call whose kernels have not been launched yet (Graph lazy batching, SURVEY 8f next-2 behind the UNMODIFIED
    losses): an EasyDict that launches on first READ.  `corres_loss.py:158-166` issues the two correspondence renders back to back and
    reads neither before both exist; deferring the launch lets them run as ONE `render_batch` (each network once over both pixel lists)
    without touching the loss code.  Writes (`ret.ray_idx = ...`, renderer.py:188) do not launch anything."""

    def __init__(self, batch):
        super().__init__()
        object.__setattr__(self, "_lz", batch)

    def _force(self):
        b = self.__dict__.get("_lz")
        if b is not None:
            b.flush()                            # (fills every result of the batch and detaches them from it; raises if the launch fails or failed)

    def _fill(self, pred):
        object.__setattr__(self, "_lz", None)
        for k, v in pred.items():
            if not dict.__contains__(self, k):
                dict.__setitem__(self, k, v)

    def __getitem__(self, k):
        self._force()
        return dict.__getitem__(self, k)

    def __getattr__(self, k):
        if k.startswith("__"):                   # copy / pickle protocol probes: not a read of a result
            raise AttributeError(k)
        self._force()
        try:
            return dict.__getitem__(self, k)
        except KeyError:
            raise AttributeError(k)

    def __contains__(self, k):
        self._force()
        return dict.__contains__(self, k)

    def __iter__(self):
        self._force()
        return dict.__iter__(self)

    def __len__(self):
        self._force()
        return dict.__len__(self)

    def keys(self):
        self._force()
        return dict.keys(self)

    def values(self):
        self._force()
        return dict.values(self)

    def items(self):
        self._force()
        return dict.items(self)

    def get(self, k, default=None):
        self._force()
        return dict.get(self, k, default)

    def pop(self, k, *default):
        self._force()
        return dict.pop(self, k, *default)

    def copy(self):
        self._force()
        return edict(dict.copy(self))

    # copies and pickles are plain EasyDicts of the RENDERED result (a copy bound to the batch would never be filled)
    def __copy__(self):
        return self.copy()

    def __deepcopy__(self, memo):
        import copy as _copy
        self._force()
        return edict({k: _copy.deepcopy(v, memo) for k, v in dict.items(self)})

    def __reduce__(self):
        self._force()
        return (edict, (dict(dict.items(self)),))

    def __repr__(self):
        self._force()
        return dict.__repr__(self)


class _LazyBatch:
    """render calls of one iteration that were issued back to back and not read yet"""

    def __init__(self, graph, opt, iter, grad):
        self.graph, self.opt, self.iter, self.grad = graph, opt, iter, grad
        self.requests, self.results, self.done, self.error = [], [], False, None

    def compatible(self, opt, iter, grad):
        return (not self.done) and opt is self.opt and iter == self.iter and grad == self.grad and len(self.requests) < L.MAX_SEGMENTS

    def add(self, q):
        res = PendingRender(self)
        self.requests.append(q)
        self.results.append(res)
        return res

    def flush(self):
        if self.done:
            if self.error is not None:       # a later read of a batch whose launch failed: say so (the results were never filled)
                raise L.SparfError(f"the deferred render call failed when it was launched: {self.error!r}")
            return
        self.done = True
        g = self.graph
        if g._pending is self:
            g._pending = None
        try:
            with torch.set_grad_enabled(self.grad):
                if len(self.requests) == 1:          # nothing to batch: the eager route (one autograd node), on the draws taken at call time
                    q = self.requests[0]
                    preds = [g._render_now(self.opt, q["pose"], q["H"], q["W"], q["intr"], pixels=q["pixels"], ray_idx=q["ray_idx"],
                                           depth_range=q["depth_range"], iter=self.iter, mode=q["mode"], draws=q["_draws"])]
                else:
                    preds = g.render_batch(self.opt, self.requests, iter=self.iter)
        except Exception as exc:
            self.error = exc
            raise
        for res, pred in zip(self.results, preds):
            res._fill(pred)
        g.lazy_stats["batches"] += 1
        g.lazy_stats["requests"] += len(self.requests)


_LIVE_GRAPHS = None


def flush_all_pending():
    """every Graph's deferred render calls, now (hooked in front of every optimiser step: a deferred call must see the weights of the
    iteration that issued it)"""
    if _LIVE_GRAPHS is not None:
        for g in list(_LIVE_GRAPHS):
            g.flush_pending()


def _register_graph(g):
    global _LIVE_GRAPHS
    if _LIVE_GRAPHS is None:
        import weakref
        _LIVE_GRAPHS = weakref.WeakSet()
        try:
            from torch.optim.optimizer import register_optimizer_step_pre_hook
            register_optimizer_step_pre_hook(lambda *a, **k: flush_all_pending())
        except ImportError:          # (older torch: Graph.flush_pending() is the caller's to call before stepping)
            pass
    _LIVE_GRAPHS.add(g)


class Graph(torch.nn.Module):
    """NeRF model: MLP prediction + volumetric rendering."""

    def __init__(self, opt, device):
        super().__init__()
        self.opt = opt
        self.device = device
        self._pinned, self._pinned_i = {}, 0
        self._pending, self.lazy_stats = None, dict(batches=0, requests=0)          # lazy batching of back-to-back render calls (PendingRender)
        import os
        self._lazy_env = os.environ.get("SPARF_LAZY_BATCH")                          # "0" / "1" overrides opt.hip.lazy_batch (A/B runs of an unmodified trainer)
        _register_graph(self)
        self.define_renderer(opt)
        # which arithmetic an unmodified trainer got: once per process and mode, at INFO level
        from .frequency_nerf import precision_name
        name = precision_name(opt)
        if name not in _LOGGED_MODES:
            _LOGGED_MODES.add(name)
            logging.getLogger("sparf_amd").info("Graph: HIP renderer in precision mode %r (opt.hip.precision / $SPARF_PRECISION; default %r)", name,
                                                "bf16x3")

    def __getstate__(self):
        """copy.deepcopy / pickle: without the open batch of deferred render calls (it holds autograd tensors of the iteration that
        issued them) and without the pinned staging buffers and their events"""
        state = dict(self.__dict__)
        state.update(_pending=None, lazy_stats=dict(batches=0, requests=0), _pinned={}, _pinned_i=0)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        _register_graph(self)

    def define_renderer(self, opt):
        self.nerf = NeRF(opt).to(self.device)
        if opt.nerf.fine_sampling:
            self.nerf_fine = NeRF(opt, is_fine_network=True).to(self.device)
        self.embedder_pts = FrequencyEmbedder(self.opt)
        self.embedder_view = FrequencyEmbedder(self.opt)

    def re_initialize(self):
        self.nerf.initialize()
        if self.opt.nerf.fine_sampling:
            self.nerf_fine.initialize()

    def get_network_components(self):
        return [self.nerf, self.nerf_fine] if self.opt.nerf.fine_sampling else [self.nerf]

    def L1_loss(self, pred, label):
        return (pred.contiguous() - label).abs().mean()

    def MSE_loss(self, pred, label, mask=None):
        loss = (pred.contiguous() - label) ** 2
        return (loss[mask] if mask is not None else loss).mean()

    # poses: fixed ground truth here; the joint pose/NeRF trainers subclass and override
    def get_w2c_pose(self, opt, data_dict, mode=None):
        return data_dict.pose

    def get_pose(self, opt, data_dict, mode=None):
        return self.get_w2c_pose(opt, data_dict, mode)

    def get_c2w_pose(self, opt, data_dict, mode=None):
        return camera.invert_pose(self.get_w2c_pose(opt, data_dict, mode))

    # ------------------------------------------------------------------ helpers
    def _depth_range(self, opt, data_dict):
        return opt.nerf.depth.range if opt.nerf.depth.param == "inverse" else data_dict.depth_range[0]

    def _range(self, depth_range):
        """Depth range for the sampling kernels: (dmin, dmax, scale, range_dev).

        Trainers pass `data_dict.depth_range[0]`, a DEVICE tensor (renderer.py:97-108): it is
        handed to the kernels as a pointer (`range_dev`, the floats are then ignored), which
        subtract max - min in fp32 exactly as torch does for tensors -- no readback, no cache
        that a recycled allocation could poison.  Python numbers (opt.nerf.depth.range, tests)
        follow torch's scalar semantics: the subtraction in double, then one cast."""
        lo, hi = depth_range[0], depth_range[1]
        if torch.is_tensor(lo) or torch.is_tensor(hi):
            dev = torch.device(self.device)
            on_dev = [x for x in (lo, hi) if torch.is_tensor(x) and x.device.type == "cuda"]
            if on_dev:
                if torch.is_tensor(depth_range) and depth_range.numel() == 2 and depth_range.device.type == "cuda":
                    rd = depth_range.detach().reshape(2).to(device=dev, dtype=torch.float32).contiguous()
                else:
                    rd = torch.stack([torch.as_tensor(x, dtype=torch.float32, device=dev).detach().reshape(()) for x in (lo, hi)])
                return 0.0, 0.0, 0.0, rd
            flo, fhi = np.float32(_as_float(lo)), np.float32(_as_float(hi))          # host tensors: fp32 arithmetic, no sync involved
            return float(flo), float(fhi), float(np.float32(fhi - flo)), None
        return float(lo), float(hi), float(np.float32(float(hi) - float(lo))), None

    def _fine_gated_off(self, opt, iter):
        r = getattr(opt.nerf, "ratio_start_fine_sampling_at_x", None) if not hasattr(opt.nerf, "get") \
            else opt.nerf.get("ratio_start_fine_sampling_at_x", None)
        return r is not None and iter is not None and iter < opt.max_iter * r

    def _rays(self, opt, pose, H, W, intr, pixels, ray_idx):
        """Ray origins / directions of the selected pixels (renderer.py:273-291).  One fused
        launch (ops.RayGen, with backward to the pose) unless the intrinsics need a gradient or
        `opt.hip.fused_rays` is False, in which case the PyTorch restatement in camera.py runs."""
        hip = opt.get("hip", None) if hasattr(opt, "get") else getattr(opt, "hip", None)
        fused = (hip is None or hip.get("fused_rays", True)) and not intr.requires_grad
        if ray_idx is not None and ray_idx.dim() == 2 and ray_idx.shape[0] != len(pose):
            ray_idx = ray_idx.reshape(-1)
        if fused:
            if pixels is None and ray_idx is None:
                ray_idx = torch.arange(H * W, device=pose.device)
            center, ray = ops.ray_gen(pose, intr, pixels=pixels, ray_idx=None if pixels is not None else ray_idx, width=W)
        elif pixels is not None:
            center, ray = camera.get_center_and_ray_at_pixels(pose, pixels, intr=intr)
        else:
            center, ray = camera.get_center_and_ray(pose, H, W, intr=intr, ray_idx=ray_idx)
        if opt.camera.ndc:
            raise NotImplementedError("camera.ndc: the reference calls convert_NDC with a stale signature "
                                      "(renderer.py:295 vs camera.py:439); the path is dead there and unsupported here")
        return center, ray

    # ------------------------------------------------------------------ entry points
    def forward(self, opt, data_dict, iter, img_idx=None, mode=None):
        """Render a random subset of pixels (train / test-optim) or all pixels of every
        image of `data_dict` (renderer.py:77-140)."""
        batch_size = len(data_dict.idx)
        pose = self.get_w2c_pose(opt, data_dict, mode=mode)
        H, W = data_dict.image.shape[-2:]
        depth_range = self._depth_range(opt, data_dict)
        if img_idx is not None:
            ray_idx = None
            n_img = len(img_idx) if isinstance(img_idx, list) else 1
            if opt.nerf.rand_rays and mode in ["train", "test-optim"]:
                ray_idx = torch.randperm(H * W, device=self.device)[:opt.nerf.rand_rays // n_img]
            # (the reference passes img_idx into the `iter` slot here, renderer.py:117; kept keyword-correct)
            ret = self.render_image_at_specific_rays(opt, data_dict, iter, img_idx=img_idx, ray_idx=ray_idx, mode=mode or "train")
            if ray_idx is not None:
                ret.ray_idx = ray_idx
            ret.idx_img_rendered = img_idx
            return ret
        if opt.nerf.rand_rays and mode in ["train", "test-optim"]:
            ray_idx = torch.randperm(H * W, device=self.device)[:opt.nerf.rand_rays // batch_size]
            ret = self.render(opt, pose, intr=data_dict.intr, ray_idx=ray_idx, mode=mode, H=H, W=W, depth_range=depth_range, iter=iter)
            ret.ray_idx = ray_idx
        elif opt.nerf.rand_rays:
            ret = self.render_by_slices(opt, pose, intr=data_dict.intr, mode=mode, H=H, W=W, depth_range=depth_range, iter=iter)
        else:
            ret = self.render(opt, pose, intr=data_dict.intr, mode=mode, H=H, W=W, depth_range=depth_range, iter=iter)
        ret.idx_img_rendered = torch.arange(start=0, end=batch_size).to(self.device)
        return ret

    def render_image_at_specific_pose_and_rays(self, opt, data_dict, pose, intr, H, W, iter, pixels=None, ray_idx=None, mode='train'):
        """renderer.py:142-190."""
        pose = pose.unsqueeze(0) if pose.dim() == 2 else pose
        intr = intr.unsqueeze(0) if intr.dim() == 2 else intr
        depth_range = self._depth_range(opt, data_dict)
        if ray_idx is None and pixels is None:
            if opt.nerf.rand_rays:
                return self.render_by_slices(opt, pose, intr=intr, mode=mode, H=H, W=W, depth_range=depth_range, iter=iter)
            return self.render(opt, pose, intr=intr, mode=mode, H=H, W=W, depth_range=depth_range, iter=iter)
        ret = self._render_deferred(opt, pose, H, W, intr, pixels, ray_idx, depth_range, iter, mode)
        if ret is None:
            ret = self.render(opt, pose, intr=intr, pixels=pixels, ray_idx=ray_idx, mode=mode, H=H, W=W, depth_range=depth_range, iter=iter)
        ret.ray_idx = ray_idx
        return ret

    # ------------------------------------------------------------------ lazy batching of back-to-back render calls
    def flush_pending(self):
        """Launch the render calls that were deferred (PendingRender) and not read yet.  Called by every other entry point of the
        renderer and in front of every optimiser step; public for callers that change weights behind torch's back."""
        if self._pending is not None:
            self._pending.flush()

    def _render_deferred(self, opt, pose, H, W, intr, pixels, ray_idx, depth_range, iter, mode):
        """-> a PendingRender (the call joins the open batch of this iteration, or opens one), or None when the call must run now.
        Deferred: train-mode calls under autograd on explicit pixel / ray lists whose render would take the fused route
        (`opt.hip.lazy_batch`, default on).  EVERY random draw of the call is taken NOW, in the order the eager call takes them
        (`_draw_randoms`): the RNG streams -- and a test harness that injects draws around the call -- see exactly the eager sequence; only
        the kernel launches wait.  A batch is launched by the first read of any of its results, by the next call into the renderer
        that is not deferred, or by the next optimiser step."""
        hip = opt.get("hip", None) if hasattr(opt, "get") else getattr(opt, "hip", None)
        lazy = (hip is None or hip.get("lazy_batch", True)) if self._lazy_env is None else self._lazy_env != "0"
        if not lazy or (hip is not None and (not hip.get("fused_render", True) or not hip.get("fused_rays", True))) \
                or mode != "train" or not torch.is_grad_enabled() or opt.camera.ndc or intr.requires_grad:
            return None
        L.require_gpu(pose.device)
        B = pose.shape[0]
        sel = pixels if pixels is not None else ray_idx
        R = sel.shape[-2] if pixels is not None else (sel.numel() if (sel.dim() == 2 and sel.shape[0] != B) else sel.shape[-1])
        Nc, Nf = int(opt.nerf.sample_intvs), int(opt.nerf.sample_intvs_fine or 0)
        fine = bool(opt.nerf.fine_sampling) and not self._fine_gated_off(opt, iter)
        n = B * R
        prec, far = pass_precision(opt, Nc)
        rows = n * (Nc + (Nf if fine else 0))
        if n == 0 or rows > max_rows_per_call(prec, pose.device, need=rows, far=far):
            return None
        if self._pending is not None and not self._pending.compatible(opt, iter, True):
            self.flush_pending()
        if self._pending is None:
            self._pending = _LazyBatch(self, opt, iter, True)
        q = dict(pose=pose, H=H, W=W, intr=intr, pixels=pixels, ray_idx=ray_idx, depth_range=depth_range, mode=mode,
                 _draws=self._draw_randoms(opt, B, R, mode, fine))
        return self._pending.add(q)

    def _draw_randoms(self, opt, B, R, mode, fine):
        """the random draws of one `render` call, in the reference's order (renderer.py:405-407 stratified jitter; frequency_nerf.py:191-192
        coarse density noise; renderer.py:439 the fine grid; the fine pass's density noise): shared by the eager fused route and the
        deferred one"""
        Nc, Nf = int(opt.nerf.sample_intvs), int(opt.nerf.sample_intvs_fine or 0)
        dev, n = self.device, B * R
        jitter = None
        if opt.nerf.sample_stratified and mode not in ['val', 'eval', 'test']:
            jitter = torch.rand(B, R, Nc, 1, device=self.device)
            if jitter.dtype is not torch.float32 or not jitter.is_contiguous():
                jitter = jitter.float().contiguous()
        use_noise = bool(opt.nerf.density_noise_reg) and mode == "train"
        noise_c = torch.randn(n, Nc, device=dev) if use_noise else None                 # frequency_nerf.py:192
        u_mid = noise_f = None
        if fine:
            det = mode not in ['train', 'test-optim'] or (not opt.nerf.sample_stratified)
            u_mid = self._grid_midpoints_fused(Nf, det)
            noise_f = torch.randn(n, Nc + Nf, device=dev) if use_noise else None
        for z in (noise_c, noise_f):
            if z is not None and (z.dtype is not torch.float32 or not z.is_contiguous()):
                raise L.SparfError("density noise must be dense float32")
        return dict(jitter=jitter, noise_c=noise_c, u_mid=u_mid, noise_f=noise_f, use_noise=use_noise)

    def render_image_at_specific_rays(self, opt, data_dict, iter, img_idx=None, pixels=None, ray_idx=None, mode='train'):
        """renderer.py:192-248."""
        pose = self.get_w2c_pose(opt, data_dict, mode=mode)
        intr = data_dict.intr
        batch_size = pose.shape[0]
        if img_idx is not None:
            if isinstance(img_idx, (tuple, list)):
                pose, intr = pose[img_idx].view(-1, 3, 4), intr[img_idx].view(-1, 3, 3)
            else:
                pose, intr = pose[img_idx].unsqueeze(0), intr[img_idx].unsqueeze(0)
                img_idx = [img_idx]
        H, W = data_dict.image.shape[-2:]
        depth_range = self._depth_range(opt, data_dict)
        if ray_idx is None and pixels is None:
            if opt.nerf.rand_rays:
                ret = self.render_by_slices(opt, pose, intr=intr, mode=mode, H=H, W=W, depth_range=depth_range, iter=iter)
            else:
                ret = self.render(opt, pose, intr=intr, mode=mode, H=H, W=W, depth_range=depth_range, iter=iter)
        else:
            ret = self.render(opt, pose, intr=intr, pixels=pixels, ray_idx=ray_idx, mode=mode, H=H, W=W, depth_range=depth_range, iter=iter)
            ret.ray_idx = ray_idx
        ret.idx_img_rendered = torch.from_numpy(np.array(img_idx)).to(self.device) if img_idx is not None else \
            torch.arange(start=0, end=batch_size).to(self.device)
        return ret

    # ------------------------------------------------------------------ the hot function
    def render(self, opt, pose, H, W, intr, pixels=None, ray_idx=None, depth_range=None, iter=None, mode=None):
        """renderer.py:250-345: coarse pass, then (unless gated off) inverse-CDF resampling,
        sort, fine pass.  Returns an EasyDict with the reference's keys."""
        return self._render_now(opt, pose, H, W, intr, pixels, ray_idx, depth_range, iter, mode)

    def _render_now(self, opt, pose, H, W, intr, pixels=None, ray_idx=None, depth_range=None, iter=None, mode=None, draws=None):
        """`render`; draws: the call's random draws if they were taken earlier (a deferred call, `_render_deferred`)"""
        L.require_gpu(pose.device)
        self.flush_pending()
        center, ray = self._rays(opt, pose, H, W, intr, pixels, ray_idx)
        B, R = ray.shape[:2]
        Nc = opt.nerf.sample_intvs
        fused = self._render_fused(opt, center, ray, depth_range, iter, mode, draws)
        if fused is not None:
            return fused
        if draws is not None:
            raise L.SparfError("a deferred render call must take the fused route (checked when it was deferred)")
        pred = edict(origins=center, viewdirs=ray)
        depth_samples = self.sample_depth(opt, B, num_rays=R, n_samples=Nc, H=H, W=W, depth_range=depth_range, mode=mode)
        # (n_coarse: the stratified samples of sample_depth sit at the end of every ray, in the coarse pass and -- the fine samples
        # all lie below them for inverse depth, renderer.py:446 -- after the merge: what pass_precision routes to fp32 there)
        coarse = self.nerf.render_pass(opt, center, ray, depth_samples, mode=mode, n_coarse=Nc)
        coarse["t"] = depth_samples
        pred.update(coarse)
        if opt.nerf.fine_sampling and not self._fine_gated_off(opt, iter):
            Nf = opt.nerf.sample_intvs_fine
            det = mode not in ['train', 'test-optim'] or (not opt.nerf.sample_stratified)
            dmin, dmax, _, rd = self._range(depth_range)
            with torch.no_grad():
                u_mid = self._grid_midpoints(Nf, det)
                merged, _ = ops.sample_fine(coarse["weights"].view(B * R, Nc), depth_samples.view(B * R, Nc), u_mid, dmin, dmax, range_dev=rd)
            depth_all = merged.view(B, R, Nc + Nf, 1)
            fine = self.nerf_fine.render_pass(opt, center, ray, depth_all, mode=mode, n_coarse=Nc)
            fine["t"] = depth_all
            pred.update({k + "_fine": v for k, v in fine.items()})
        return pred

    def _render_fused(self, opt, center, ray, depth_range, iter, mode, draws=None):
        """The body of `render` as ONE autograd node (ops.RenderFn): coarse depths, coarse pass, resampling + merge, fine pass issued
        from one frame, all fp32 results in one allocation.  Same kernels, same draws in the same order (jitter, coarse density noise,
        fine grid, fine density noise: renderer.py:405-407, frequency_nerf.py:191-192, renderer.py:439), same results bit for bit as
        the pass-by-pass path below it (tests/test_graph_gpu.py::test_fused_render_equals_pass_by_pass), which remains for renders
        larger than one launch set and for `opt.hip.fused_render = False`.  -> EasyDict, or None when the pass-by-pass path must run."""
        hip = opt.get("hip", None) if hasattr(opt, "get") else getattr(opt, "hip", None)
        if hip is not None and not hip.get("fused_render", True):
            return None
        B, R = ray.shape[:2]
        Nc, Nf = int(opt.nerf.sample_intvs), int(opt.nerf.sample_intvs_fine or 0)
        fine = bool(opt.nerf.fine_sampling) and not self._fine_gated_off(opt, iter)
        n = B * R
        prec, far = pass_precision(opt, Nc)
        if n == 0 or n * (Nc + (Nf if fine else 0)) > max_rows_per_call(prec, ray.device, need=n * (Nc + (Nf if fine else 0)), far=far):
            return None
        dev = ray.device
        dmin, dmax, scale, rd = self._range(depth_range)
        if draws is None:
            draws = self._draw_randoms(opt, B, R, mode, fine)
        jitter, noise_c, u_mid, noise_f, use_noise = draws["jitter"], draws["noise_c"], draws["u_mid"], draws["noise_f"], draws["use_noise"]
        pc = self.nerf.hip_params()
        pf = self.nerf_fine.hip_params() if fine else None
        grad = torch.is_grad_enabled()
        theta_c = self.nerf.flat_params(pc) if grad else None
        theta_f = self.nerf_fine.flat_params(pf) if (grad and fine) else None
        fprec = far[1] if far is not None else None
        cfg = dict(R=n, Nc=Nc, Nf=Nf, fine=fine, dmin=dmin, dmax=dmax, scale=scale, inverse=opt.nerf.depth.param == "inverse", u_const=0.5,
                   noise_scale=float(opt.nerf.density_noise_reg) if use_noise else 0.0, white_bg=bool(opt.nerf.setbg_opaque or opt.mask_img),
                   prec_c=prec, prec_f=prec, far_c=far, far_f=far, c2f=tuple(float(x) for x in opt.barf_c2f) if opt.barf_c2f is not None else None)
        coarse, fine_out = ops.render_fused(
            center.reshape(n, 3), ray.reshape(n, 3), cfg, jitter.view(n, Nc) if jitter is not None else None, u_mid, noise_c, noise_f, rd,
            self.nerf.packed(prec, pc), self.nerf_fine.packed(prec, pf) if fine else None,
            self.nerf.packed(fprec, pc) if far is not None else None, self.nerf_fine.packed(fprec, pf) if (fine and far is not None) else None,
            self.nerf.progress.detach(), self.nerf_fine.progress.detach() if fine else None, theta_c, theta_f)
        pred = edict(origins=center, viewdirs=ray)

        def shaped(o, N):
            return dict(rgb_samples=o["rgb_samples"].view(B, R, N, 3), density_samples=o["density_samples"].view(B, R, N), rgb=o["rgb"].view(B, R, 3),
                        rgb_var=o["rgb_var"].view(B, R, 1), depth=o["depth"].view(B, R, 1), depth_var=o["depth_var"].view(B, R, 1),
                        opacity=o["opacity"].view(B, R, 1), weights=o["weights"].view(B, R, N, 1), all_cumulated=o["all_cumulated"].view(B, R),
                        t=o["t"].view(B, R, N, 1))
        pred.update(shaped(coarse, Nc))
        if fine:
            pred.update({k + "_fine": v for k, v in shaped(fine_out, Nc + Nf).items()})
        return pred

    def render_by_slices(self, opt, pose, H, W, intr, depth_range, iter, mode=None):
        """Whole images (renderer.py:347-381).  The reference walks H*W in `rand_rays`-sized
        slices to fit 20 GB; here a slice is as large as one kernel launch allows."""
        keys = ["rgb", "rgb_var", "depth", "depth_var", "opacity", "normal", "all_cumulated"]
        ret_all = edict({k: [] for k in keys})
        if opt.nerf.fine_sampling and not self._fine_gated_off(opt, iter):
            ret_all.update({k + "_fine": [] for k in keys})
        B = len(pose)
        n_per_ray = opt.nerf.sample_intvs + (opt.nerf.sample_intvs_fine if opt.nerf.fine_sampling else 0)
        step = max(int(opt.nerf.rand_rays), max_rows_per_call(pass_precision(opt, opt.nerf.sample_intvs)[0], self.device) // max(1, B * n_per_ray))
        for c in range(0, H * W, step):
            ray_idx = torch.arange(c, min(c + step, H * W), device=self.device)
            ret = self.render(opt, pose, H=H, W=W, intr=intr, ray_idx=ray_idx, depth_range=depth_range, iter=iter, mode=mode)
            for k in ret_all:
                if k in ret.keys():
                    ret_all[k].append(ret[k])
        for k in ret_all:
            ret_all[k] = torch.cat(ret_all[k], dim=1) if len(ret_all[k]) > 0 else None
        return ret_all

    @torch.no_grad()
    def evaluate_psnr(self, opt, pose, H, W, intr, depth_range, image, iter=None, mode="val"):
        """Full-image evaluation with the squared error accumulated on the device slice by slice (the
        slices of renderer.py:347-381 + PSNR = -10 log10 MSE, metrics.py:246 / nerf_trainer.py:298-305), without keeping
        any per-ray output: one `photometric_loss` launch per slice and head.  `image`: [B, 3, H, W] or [B, H*W, 3].
        -> dict(mse, psnr[, mse_fine, psnr_fine]) of 0-d device tensors (PSNR over all B images together, like the
        reference's `MSE_loss` of the whole batch)."""
        B = len(pose)
        target = image.reshape(B, 3, H * W).permute(0, 2, 1) if image.dim() == 4 else image.reshape(B, H * W, 3)
        fine = opt.nerf.fine_sampling and not self._fine_gated_off(opt, iter)
        n_per_ray = opt.nerf.sample_intvs + (opt.nerf.sample_intvs_fine if opt.nerf.fine_sampling else 0)
        step = max(int(opt.nerf.rand_rays), max_rows_per_call(pass_precision(opt, opt.nerf.sample_intvs)[0], self.device) // max(1, B * n_per_ray))
        sq = torch.zeros(2, device=self.device, dtype=torch.float64)
        for c in range(0, H * W, step):
            hi = min(c + step, H * W)
            ray_idx = torch.arange(c, hi, device=self.device)
            ret = self.render(opt, pose, H=H, W=W, intr=intr, ray_idx=ray_idx, depth_range=depth_range, iter=iter, mode=mode)
            tgt = target[:, c:hi].contiguous()
            n = tgt.numel()
            sq[0] += ops.photometric_loss(ret.rgb, tgt).double() * n
            if fine:
                sq[1] += ops.photometric_loss(ret.rgb_fine, tgt).double() * n
        mse = (sq / (B * H * W * 3)).float()
        out = edict(mse=mse[0], psnr=-10 * mse[0].log10())
        if fine:
            out.update(mse_fine=mse[1], psnr_fine=-10 * mse[1].log10())
        return out

    # ------------------------------------------------------------------ depth sampling
    def sample_depth(self, opt, batch_size, n_samples, H, W, depth_range, num_rays=None, mode=None):
        """Stratified samples along every ray, same range for all rays (renderer.py:383-419).
        Returns [B, num_rays, n_samples, 1]."""
        return self._sample_depth(opt, batch_size, n_samples, H, W, depth_range, num_rays, mode)

    def _sample_depth(self, opt, batch_size, n_samples, H, W, depth_range, num_rays=None, mode=None, out=None, jitter=None):
        """sample_depth, optionally writing into `out` ([B*num_rays, n_samples] rows of a shared buffer, render_batch); jitter: the
        stratified draw if it was taken earlier (a deferred call, `_render_deferred`)"""
        num_rays = H * W if num_rays is None else num_rays          # (the reference's `or` maps an empty batch to H*W)
        dmin, _, scale, rd = self._range(depth_range)
        if jitter is None and opt.nerf.sample_stratified and mode not in ['val', 'eval', 'test']:
            jitter = torch.rand(batch_size, num_rays, n_samples, 1, device=self.device)
        t = ops.sample_coarse(batch_size * num_rays, n_samples, dmin, scale, opt.nerf.depth.param == "inverse", self.device,
                              jitter=jitter, u_const=0.5, range_dev=rd, out=out)
        return t.view(batch_size, num_rays, n_samples, 1)

    def _grid_midpoints_fused(self, n_fine, det):
        """the mid-points for the fused render: a HOST float32 tensor where the grid is drawn on the CPU as the reference does
        (renderer.py:439) and fits the launch arguments (C ABI sparf_sample_fine_hostgrid: no host -> device copy at all), else the
        device tensor of _grid_midpoints"""
        hip = self.opt.get("hip", None) if hasattr(self.opt, "get") else None
        if det or n_fine > 256 or (hip is not None and hip.get("device_rng", False)):
            return self._grid_midpoints(n_fine, det)
        cpu = torch.rand(n_fine + 1)
        if cpu.device.type != "cpu" or cpu.dtype is not torch.float32:
            return 0.5 * (cpu[:-1] + cpu[1:]).to(device=self.device, dtype=torch.float32)
        return (0.5 * (cpu[:-1] + cpu[1:])).contiguous()       # the same two fp32 operations the reference runs on the device: same bits

    def _grid_midpoints(self, n_fine, det):
        """mid-points u_j = (g_j + g_{j+1}) / 2 of the fine-sampling grid (renderer.py:434-441), [n_fine] on the device"""
        if det:
            key = ("det", n_fine)
            mid = self._pinned.get(key)
            if mid is None:                  # a constant of (n_fine, device)
                grid = torch.linspace(0, 1, n_fine + 1, device=self.device)
                mid = self._pinned[key] = 0.5 * (grid[:-1] + grid[1:])
            return mid
        # one shared, unsorted draw made on the CPU, as the reference does (renderer.py:439),
        # but staged through a small ring of pinned buffers: a pageable .to(device) would
        # block the host behind everything already queued on the stream, every step
        hip = self.opt.get("hip", None) if hasattr(self.opt, "get") else None
        if hip is not None and hip.get("device_rng", False):
            # opt.hip.device_rng: draw on the device instead (same distribution, different
            # RNG stream) -- keeps the whole step free of host->device copies, which is
            # what hipGraph capture of a training step needs
            grid = torch.rand(n_fine + 1, device=self.device)
            return 0.5 * (grid[:-1] + grid[1:])
        cpu = torch.rand(n_fine + 1)
        if cpu.device.type != "cpu":         # (a test harness replaying recorded draws may hand back a device tensor)
            return 0.5 * (cpu[:-1] + cpu[1:]).to(self.device)
        mid = 0.5 * (cpu[:-1] + cpu[1:])     # the same two fp32 operations the reference runs on the device: same bits
        if torch.device(self.device).type == "cuda":
            ring = self._pinned.setdefault(n_fine, [[torch.empty(n_fine).pin_memory(), None] for _ in range(8)])
            self._pinned_i = (self._pinned_i + 1) % len(ring)
            slot = ring[self._pinned_i]
            if slot[1] is not None:
                slot[1].synchronize()      # the async copy that last read this staging buffer has finished (host > 8 renders ahead)
            out = slot[0].copy_(mid).to(self.device, non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record(torch.cuda.current_stream(self.device))
            return out
        return mid.to(self.device)

    def sample_depth_from_pdf(self, opt, weights, n_samples_coarse, n_samples_fine, depth_range, det):
        """Inverse-transform resampling of the coarse weights (renderer.py:421-456);
        weights [B, num_rays, Nc] -> [B, num_rays, Nf, 1] (unsorted, as the reference)."""
        B, R = weights.shape[:2]
        dmin, dmax, _, rd = self._range(depth_range)
        dummy_t = torch.zeros(B * R, n_samples_coarse, device=weights.device)
        _, tf = ops.sample_fine(weights.reshape(B * R, n_samples_coarse), dummy_t, self._grid_midpoints(n_samples_fine, det),
                                dmin, dmax, want_unsorted=True, range_dev=rd)
        return tf.view(B, R, n_samples_fine, 1)

    # ------------------------------------------------------------------ render up to a per-ray depth
    def render_up_to_maxdepth_at_specific_pose_and_rays(self, opt, data_dict, pose, intr, H, W, depth_max, iter,
                                                        pixels=None, ray_idx=None, mode='train'):
        """renderer.py:460-502."""
        pose = pose.unsqueeze(0) if pose.dim() == 2 else pose
        intr = intr.unsqueeze(0) if intr.dim() == 2 else intr
        depth_range = self._depth_range(opt, data_dict)
        ret = self.render_to_max(opt, pose, intr=intr, pixels=pixels, ray_idx=ray_idx, mode=mode, H=H, W=W,
                                 depth_min=depth_range[0], depth_max=depth_max, iter=iter)
        ret.ray_idx = ray_idx
        return ret

    def render_to_max(self, opt, pose, H, W, intr, pixels=None, ray_idx=None, depth_max=None, depth_min=None, iter=None, mode=None):
        """renderer.py:504-593: deterministic samples up to a per-ray far bound; the fine
        network is evaluated on the SAME samples (:583-592)."""
        L.require_gpu(pose.device)
        self.flush_pending()
        center, ray = self._rays(opt, pose, H, W, intr, pixels, ray_idx)
        B, R = ray.shape[:2]
        pred = edict(origins=center, viewdirs=ray)
        depth_samples = self.sample_depth_diff_max_range_per_ray(opt, B, num_rays=R, n_samples=opt.nerf.sample_intvs, H=H, W=W,
                                                                 depth_max=depth_max, depth_min=depth_min, mode=mode)
        coarse = self.nerf.render_pass(opt, center, ray, depth_samples, mode=mode, to_max=True)
        coarse["t"] = depth_samples
        pred.update(coarse)
        skip = self._fine_gated_off(opt, iter)
        s = getattr(opt.nerf, "start_fine_sampling_at_x", None) if not hasattr(opt.nerf, "get") else opt.nerf.get("start_fine_sampling_at_x", None)
        if not skip and s is not None and iter is not None and iter < s:
            skip = True
        if opt.nerf.fine_sampling and not skip:
            fine = self.nerf_fine.render_pass(opt, center, ray, depth_samples, mode=mode, to_max=True)
            fine["t"] = depth_samples
            pred.update({k + "_fine": v for k, v in fine.items()})
        return pred

    # ------------------------------------------------------------------ several render calls in one set of launches
    def render_batch(self, opt, requests, iter=None):
        """what the SPARF losses issue as 5-6 separate render calls per
        iteration (photometric `render`, 2 correspondence renders `corres_loss.py:158-166`,
        3 depth-consistency renders incl. `render_to_max` under no_grad
        `depth_cons_loss.py:192,267,291`) evaluated with ONE fused pass per (network, sample
        count, grad mode) group instead of one per call.

        Rays are independent, so the requests of a group share one set of ray buffers: every
        request's ray generation and depth sampling write their rows AT THE REQUEST'S OFFSET of the
        group's [2, R_total, 3] ray buffer / [R_total, N] depth buffer (ops.ray_gen(out=),
        ops.sample_coarse(out=)), the pass runs once over all of them with a segment table
        (include/sparf_hip.h sparf_segment_t: per-request noise scale, per-request upstream
        gradients), and every request's results are views of the pass's outputs: no
        concatenation, no split copies, no gradient gathers.

        requests: list of dicts with the keyword arguments of `render` (pose, H, W, intr,
        pixels | ray_idx, depth_range, mode) or, with key `depth_max`, of `render_to_max`
        (pose, H, W, intr, pixels | ray_idx, depth_min, depth_max, mode); optional
        `no_grad=True` renders that request without autograd state (inference kernels).
        A request may carry `_draws` (Graph._draw_randoms: its stratified jitter, fine grid and density noise, taken when the call was
        issued -- the deferred calls of the lazy batching); without it the draws are made here.
        Returns the list of EasyDicts the separate calls would return."""
        L.require_gpu(self.device)
        self.flush_pending()
        Nc = opt.nerf.sample_intvs
        Nf = opt.nerf.sample_intvs_fine
        reg = float(opt.nerf.density_noise_reg) if opt.nerf.density_noise_reg else 0.0
        dev = self.device
        fine_on = bool(opt.nerf.fine_sampling) and not self._fine_gated_off(opt, iter)
        s0 = opt.nerf.get("start_fine_sampling_at_x", None) if hasattr(opt.nerf, "get") else getattr(opt.nerf, "start_fine_sampling_at_x", None)
        tomax_skip = s0 is not None and iter is not None and iter < s0
        white_bg = bool(opt.nerf.setbg_opaque or opt.mask_img)

        def count(q):
            """rays of a request without generating them: B * (pixels | ray_idx rows | H*W)"""
            B = q["pose"].shape[0]
            sel = q.get("pixels") if q.get("pixels") is not None else q.get("ray_idx")
            if sel is None:
                return B, q["H"] * q["W"]
            if q.get("pixels") is None and sel.dim() == 2 and sel.shape[0] != B:
                return B, sel.numel()
            return B, sel.shape[-2] if q.get("pixels") is not None else sel.shape[-1]

        items = []
        for q in requests:
            q = dict(q)
            B, R = count(q)
            items.append(dict(q=q, mode=q.get("mode"), to_max="depth_max" in q, B=B, R=R, n=B * R,
                              nograd=bool(q.get("no_grad", False)) or not torch.is_grad_enabled()))

        # precision of the passes: `render` requests carry the stratified coarse samples at the end of every ray (far-row routing
        # under inverse depth, frequency_nerf.pass_precision), render_to_max requests do not -- where the two differ they run as
        # separate passes
        prec_r = pass_precision(opt, Nc)
        for nograd in (False, True):
            # render requests first, render_to_max requests after them: each kind is then one contiguous row range
            members = sorted([m for m in items if m["nograd"] == nograd], key=lambda m: m["to_max"])
            if not members:
                continue
            Rtot = sum(m["n"] for m in members)
            with torch.set_grad_enabled(not nograd):
                # ray generation of the whole group into one (centres, directions) buffer, each request at its offset
                hip = opt.get("hip", None) if hasattr(opt, "get") else getattr(opt, "hip", None)
                # (pixel lists that carry a gradient -- depth_cons_loss.py:291 -- take the per-request path: ops.RayGenMany sees its
                # pixel lists as plain data)
                fused = (hip is None or hip.get("fused_rays", True)) and not any(
                    m["q"]["intr"].requires_grad or (m["q"].get("pixels") is not None and m["q"]["pixels"].requires_grad) for m in members)
                off, specs = 0, []
                for m in members:
                    m["off"] = off
                    off += m["n"]
                    q = m["q"]
                    px, ix = q.get("pixels"), q.get("ray_idx")
                    if px is None and ix is None:
                        ix = torch.arange(q["H"] * q["W"], device=dev)
                    if ix is not None and ix.dim() == 2 and ix.shape[0] != m["B"]:
                        ix = ix.reshape(-1)
                    specs.append((q["intr"], px, None if px is not None else ix, q["W"]))
                if opt.camera.ndc:
                    raise NotImplementedError("camera.ndc is dead code in the reference (renderer.py:295 vs camera.py:439) and unsupported here")
                if fused:
                    rays = ops.ray_gen_many(specs, [m["q"]["pose"] for m in members])
                else:       # PyTorch ray generation (intrinsics with a gradient): the per-request results, concatenated
                    cr = [self._rays(opt, m["q"]["pose"], m["q"]["H"], m["q"]["W"], m["q"]["intr"], m["q"].get("pixels"), m["q"].get("ray_idx")) for m in members]
                    rays = torch.stack([torch.cat([c.reshape(-1, 3) for c, _ in cr]), torch.cat([r.reshape(-1, 3) for _, r in cr])])
                t_all = torch.empty(Rtot, Nc, device=dev, dtype=torch.float32)
                for m in members:                 # coarse depths, each request at its offset
                    q, n, off = m["q"], m["n"], m["off"]
                    m["pred"] = edict(origins=rays[0, off:off + n].view(m["B"], m["R"], 3), viewdirs=rays[1, off:off + n].view(m["B"], m["R"], 3))
                    tv = t_all[off:off + n]
                    if n > 0 and m["to_max"]:
                        self._sample_depth_to_max(opt, m["B"], num_rays=m["R"], n_samples=Nc, H=q["H"], W=q["W"],
                                                  depth_max=q["depth_max"], depth_min=q["depth_min"], mode=m["mode"], out=tv)
                    elif n > 0:
                        self._sample_depth(opt, m["B"], num_rays=m["R"], n_samples=Nc, H=q["H"], W=q["W"], depth_range=q["depth_range"],
                                           mode=m["mode"], out=tv, jitter=(q.get("_draws") or {}).get("jitter"))
                    m["t"] = tv.view(m["B"], m["R"], Nc, 1)

                def run(net, group, t_buf, N, key_t, suffix):
                    """one pass of `net` over the rays of `group` (contiguous members of this grad-mode block); more than
                    L.MAX_SEGMENTS requests, or more sample rows than one launch set takes, run as consecutive passes"""
                    if not group:
                        return
                    prec_m = pass_precision(opt, None, to_max_samples=N)     # (by value, under this group's grad mode)
                    kinds = {m["to_max"] for m in group}
                    if len(kinds) == 2 and prec_r != prec_m:          # render and render_to_max requests at different precisions
                        run(net, [m for m in group if not m["to_max"]], t_buf, N, key_t, suffix)
                        return run(net, [m for m in group if m["to_max"]], t_buf, N, key_t, suffix)
                    prec, far = prec_m if group[0]["to_max"] else prec_r
                    far = (far[0], far[1], net.packed(far[1])) if far is not None else None
                    cap = max_rows_per_call(prec, dev, need=sum(m["n"] for m in group) * N) // N
                    if len(group) > L.MAX_SEGMENTS or sum(m["n"] for m in group) > cap:
                        part, rows = [], 0
                        for m in group:
                            if m["n"] > cap:
                                raise L.SparfError(f"render_batch: one request of {m['n']} rays x {N} samples exceeds a launch set; render it with render()")
                            if part and (len(part) == L.MAX_SEGMENTS or rows + m["n"] > cap):
                                run(net, part, t_buf, N, key_t, suffix)
                                part, rows = [], 0
                            part.append(m)
                            rows += m["n"]
                        return run(net, part, t_buf, N, key_t, suffix)
                    lo, hi = group[0]["off"], group[-1]["off"] + group[-1]["n"]
                    segs = [(m["off"] - lo, m["n"], reg if (m["mode"] == "train" and reg > 0) else 0.0) for m in group]
                    noise = None
                    if any(s[2] > 0 for s in segs):      # frequency_nerf.py:191-192, per-request scale in the table
                        pre = [(m["q"].get("_draws") or {}).get("noise_c" if suffix == "" else "noise_f") for m in group]
                        if all(p is not None for p in pre):           # deferred calls: the draws they took when they were issued
                            noise = pre[0] if len(pre) == 1 else torch.cat([p.reshape(m["n"], N) for p, m in zip(pre, group)])
                        else:
                            noise = torch.randn(hi - lo, N, device=dev)
                    outs = ops.nerf_pass_segments(rays[0, lo:hi], rays[1, lo:hi], t_buf[lo:hi], noise, white_bg, prec, net.packed(prec),
                                                  net.band_weights(), net.hip_params(), segs, far=far)
                    for m, o in zip(group, outs):
                        B, R = m["B"], m["R"]
                        part = dict(rgb_samples=o["rgb_samples"].view(B, R, N, 3), density_samples=o["density_samples"].view(B, R, N),
                                    rgb=o["rgb"].view(B, R, 3), rgb_var=o["rgb_var"].view(B, R, 1), depth=o["depth"].view(B, R, 1),
                                    depth_var=o["depth_var"].view(B, R, 1), opacity=o["opacity"].view(B, R, 1),
                                    weights=o["weights"].view(B, R, N, 1), all_cumulated=o["all_cumulated"].view(B, R), t=m[key_t])
                        m["out" + suffix] = part
                        m["pred"].update({k + suffix: v for k, v in part.items()})

                run(self.nerf, members, t_all, Nc, "t", "")
                if fine_on:
                    # render requests: resample + merge per request (own depth range / grid), into one [R, Nc+Nf] buffer; they are
                    # laid first, render_to_max requests (fine network on the SAME samples, renderer.py:583-592) after them
                    rend = [m for m in members if not m["to_max"]]
                    tomx = [m for m in members if m["to_max"]]
                    if rend:
                        lo = rend[0]["off"]
                        t_fine = torch.empty(sum(m["n"] for m in rend), Nc + Nf, device=dev, dtype=torch.float32)
                        for m in rend:
                            det = m["mode"] not in ['train', 'test-optim'] or (not opt.nerf.sample_stratified)
                            dmin, dmax, _, rd = self._range(m["q"]["depth_range"])
                            tv = t_fine[m["off"] - lo:m["off"] - lo + m["n"]]
                            if m["n"] > 0:
                                u_pre = (m["q"].get("_draws") or {}).get("u_mid")
                                with torch.no_grad():
                                    ops.sample_fine(m["out"]["weights"].reshape(m["n"], Nc), m["t"].reshape(m["n"], Nc),
                                                    u_pre if u_pre is not None else self._grid_midpoints(Nf, det), dmin, dmax, range_dev=rd, out=tv)
                            m["t_fine"] = tv.view(m["B"], m["R"], Nc + Nf, 1)
                        run(self.nerf_fine, rend, _Shifted(t_fine, lo), Nc + Nf, "t_fine", "_fine")
                    if tomx and not tomax_skip:
                        run(self.nerf_fine, tomx, t_all, Nc, "t", "_fine")
        return [m["pred"] for m in items]

    def sample_depth_diff_max_range_per_ray(self, opt, batch_size, n_samples, H, W, depth_min, depth_max, num_rays=None, mode=None):
        """t_i = (i+1)/n * (depth_max[b,r] - depth_min) + depth_min (renderer.py:595-624); metric only."""
        return self._sample_depth_to_max(opt, batch_size, n_samples, H, W, depth_min, depth_max, num_rays, mode)

    def _sample_depth_to_max(self, opt, batch_size, n_samples, H, W, depth_min, depth_max, num_rays=None, mode=None, out=None):
        num_rays = H * W if num_rays is None else num_rays          # (the reference's `or` maps an empty batch to H*W)
        if torch.is_tensor(depth_min) and depth_min.device.type == "cuda":      # data_dict.depth_range[0][0]: stays on the device
            dmin, rd = 0.0, depth_min.detach().reshape(-1)[:1].to(device=self.device, dtype=torch.float32).contiguous()
        else:
            dmin, rd = float(np.float32(_as_float(depth_min))), None
        t = ops.sample_coarse(batch_size * num_rays, n_samples, dmin, 0.0, False, self.device, u_const=1.0,
                              dmax_ray=depth_max.reshape(batch_size * num_rays), range_dev=rd, out=out)
        return t.view(batch_size, num_rays, n_samples, 1)
