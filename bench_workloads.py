"""Synthetic workloads of bench.py: one training iteration of each BASELINE.json config (1-4) on
top of the drop-in `Graph`, with the shapes and the call sequence of the reference trainers.

The trainers and losses themselves are out of scope: what is restated
here, in plain PyTorch host code, is only what is needed to DRIVE the renderer the way they do
-- which render methods are called, with which pose / pixel / ray shapes, under which grad
mode -- and to put a gradient on the outputs they differentiate:

  config 1  nerf_training_w_gt_poses/dtu/nerf.py            fixed poses, photometric loss on 4 x 1024 rays
  config 2  joint_pose_nerf_training/dtu/barf.py            3 noisy views, SE(3) refinement parameters behind
            get_w2c_pose (joint_pose_nerf_trainer.py:710-749), BARF c2f driven through progress.data.fill_
            (nerf_trainer.py:273-275), Graph.forward -> randperm rays (renderer.py:77-140)
  config 3  joint_pose_nerf_training/llff/sparf.py          inverse depth [1,0]; per iteration the SPARF call mix:
            photometric forward + 2 correspondence renders on pixel lists (corres_loss.py:158-166) + 3
            depth-consistency renders (depth_cons_loss.py:192, :267 render_up_to_maxdepth under no_grad, :291)
  config 4  joint_pose_nerf_training/replica/sparf.py       the same mix on 9 views, 340x600, metric depth [0.1, 6.5]
"""
import contextlib
import math

import numpy as np
import torch

from sparf_amd import ops
from sparf_amd.config import default_opt, _merge
from sparf_amd.edict import EasyDict as edict
from sparf_amd.renderer import Graph

SHAPES = {
    1: dict(B=4, H=300, W=400, f=500.0, rng=(1.2, 5.2), layout="ring", what="DTU-shaped scene, fixed GT poses, photometric loss"),
    2: dict(B=3, H=300, W=400, f=500.0, rng=(1.2 * 0.8, 5.2 * 1.2), layout="ring",
            what="DTU-shaped scene, 3 noisy views, joint pose-NeRF step (BARF c2f + SE(3) refinement), photometric loss"),
    3: dict(B=3, H=378, W=504, f=420.0, rng=(1, 0), layout="forward",
            what="LLFF-shaped scene (378x504, inverse depth [1,0]), SPARF call mix: photometric + 2 correspondence + 3 depth-consistency renders"),
    4: dict(B=9, H=340, W=600, f=300.0, rng=(0.1, 6.5), layout="ring",
            what="Replica-shaped scene (340x600, 9 views, depth [0.1,6.5]), joint pose-NeRF + SPARF call mix"),
}


def config_opt(config, precision, rays=4096, **over):
    """Options of the reference settings file of each config, with BASELINE's 64 + 128 samples."""
    base = dict(nerf=dict(fine_sampling=True, sample_intvs=64, sample_intvs_fine=128, rand_rays=rays, depth=dict(param="metric")),
                hip=dict(precision=precision))
    if config == 1:
        base["nerf"]["density_noise_reg"] = True              # nerf_training_w_gt_poses/dtu/nerf.py:34
    else:
        base["barf_c2f"] = [0.4, 0.7]                         # dtu/barf.py:36, llff/sparf.py:38, replica/sparf.py:38
    if config == 3:
        base["nerf"]["depth"] = dict(param="inverse", range=[1, 0])
    if config == 4:
        base["nerf"]["ratio_start_fine_sampling_at_x"] = 0.25
    o = default_opt(**base)
    _merge(o, over)
    return o


def cameras(config, device, seed=0):
    """B world-to-camera poses + intrinsics: a ring looking at the origin (DTU / Replica-like) or a
    small forward-facing rig (LLFF-like)."""
    s = SHAPES[config]
    B, H, W = s["B"], s["H"], s["W"]
    poses = []
    for b in range(B):
        if s["layout"] == "ring":
            ang = 2 * math.pi * b / B
            c = torch.tensor([3.2 * math.cos(ang), 0.4, 3.2 * math.sin(ang)])
            z = -c / c.norm()
        else:
            ang = 2 * math.pi * b / B
            c = torch.tensor([0.25 * math.cos(ang), 0.25 * math.sin(ang), 0.0])
            tgt = torch.tensor([0.0, 0.0, 4.0])
            z = (tgt - c) / (tgt - c).norm()
        x = torch.linalg.cross(torch.tensor([0.0, 1.0, 0.0]), z)
        x = x / x.norm()
        y = torch.linalg.cross(z, x)
        R_w2c = torch.stack([x, y, z], dim=1).T
        poses.append(torch.cat([R_w2c, (-R_w2c @ c)[:, None]], dim=1))
    pose = torch.stack(poses).to(device)
    intr = torch.tensor([[s["f"], 0, W / 2], [0, s["f"], H / 2], [0, 0, 1]]).repeat(B, 1, 1).to(device)
    return pose, intr


def analytic_images(pose, intr, H, W, centre=(0.0, 0.0, 0.0)):
    """Closed-form target images of a multi-view consistent scene: a unit sphere at `centre` whose
    colour is a smooth function of the surface point, in front of a vertical colour gradient.
    pose [B,3,4] w2c -> images [B,3,H,W]."""
    dev = pose.device
    B = pose.shape[0]
    ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32) + 0.5, torch.arange(W, device=dev, dtype=torch.float32) + 0.5, indexing="ij")
    pix = torch.stack([xs, ys, torch.ones_like(xs)], dim=-1).reshape(1, H * W, 3)
    cam = pix @ torch.linalg.inv(intr).transpose(-1, -2)
    R, t = pose[:, :, :3], pose[:, :, 3]
    o = -(R.transpose(-1, -2) @ t[..., None])[..., 0] - torch.tensor(centre, device=dev)     # camera centres relative to the sphere [B,3]
    d = cam @ R                                                          # R^T applied to row vectors
    d = d / d.norm(dim=-1, keepdim=True)
    bq = (d * o[:, None]).sum(-1)
    cq = (o * o).sum(-1)[:, None] - 1.0
    disc = bq * bq - cq
    hit = disc > 0
    tt = -bq - disc.clamp(min=0).sqrt()
    p = o[:, None] + d * tt[..., None]
    col_s = 0.5 + 0.5 * torch.stack([torch.sin(3 * p[..., 0]), torch.sin(3 * p[..., 1] + 1.0), torch.sin(3 * p[..., 2] + 2.0)], dim=-1)
    v = 0.5 + 0.5 * d[..., 1:2]
    col_b = torch.cat([0.2 + 0.6 * v, 0.3 + 0.3 * v, 0.8 - 0.5 * v], dim=-1)
    col = torch.where((hit & (tt > 0))[..., None], col_s, col_b)
    return col.reshape(B, H, W, 3).permute(0, 3, 1, 2).contiguous()


def se3_exp(xi):
    """xi [B,6] = (rotation vector w, translation u) -> [B,3,4] rigid transforms [R | V u]: the closed form
    of the twist exponential, as the reference's camera.lie.se3_to_SE3 (source/utils/camera.py) writes it --
    R = I + A wx + B wx^2, V = I + B wx + C wx^2 with A = sin(t)/t, B = (1-cos t)/t^2, C = (t-sin t)/t^3
    (Taylor series near t = 0).  A dozen small tensor ops; torch.matrix_exp costs ~150 launches fwd + bwd."""
    w, u = xi[:, :3], xi[:, 3:]
    t2 = (w * w).sum(-1)
    small = t2 < 1e-8
    t2s = torch.where(small, torch.ones_like(t2), t2)
    t = t2s.sqrt()
    A = torch.where(small, 1 - t2 / 6, torch.sin(t) / t)
    Bc = torch.where(small, 0.5 - t2 / 24, (1 - torch.cos(t)) / t2s)
    C = torch.where(small, 1.0 / 6 - t2 / 120, (t - torch.sin(t)) / (t2s * t))
    O = torch.zeros_like(t2)
    wx = torch.stack([O, -w[:, 2], w[:, 1], w[:, 2], O, -w[:, 0], -w[:, 1], w[:, 0], O], dim=-1).view(-1, 3, 3)
    wx2 = wx @ wx
    I = torch.eye(3, device=xi.device, dtype=xi.dtype)
    R = I + A[:, None, None] * wx + Bc[:, None, None] * wx2
    V = I + Bc[:, None, None] * wx + C[:, None, None] * wx2
    return torch.cat([R, V @ u[:, :, None]], dim=-1)


def compose(a, b):
    """a o b for [B,3,4] rigid transforms: x -> a(b(x))"""
    return torch.cat([a[:, :, :3] @ b[:, :, :3], a[:, :, :3] @ b[:, :, 3:] + a[:, :, 3:]], dim=-1)


class PoseGraph(Graph):
    """What PoseAndNerfTrainerPerScene builds (joint_pose_nerf_trainer.py:710-749): the renderer
    plus learnable pose refinements behind get_w2c_pose."""

    def __init__(self, opt, device, init_pose):
        super().__init__(opt, device)
        self.init_pose = init_pose.to(device)
        self.se3_refine = torch.nn.Parameter(torch.zeros(len(init_pose), 6, device=device))

    def get_w2c_pose(self, opt, data_dict, mode=None):
        return compose(se3_exp(self.se3_refine), self.init_pose)


_BOTTOM = {}


def to44(p):
    key = (p.device, p.dtype)
    if key not in _BOTTOM:           # a constant of the device: built once (a `torch.tensor([...], device=...)` per call is a host -> device copy per call)
        _BOTTOM[key] = torch.tensor([0.0, 0.0, 0.0, 1.0], device=p.device, dtype=p.dtype)
    return torch.cat([p, _BOTTOM[key].expand(*p.shape[:-2], 1, 4)], dim=-2)


def project(px, depth, K_i, K_j, T_ij):
    """batched_geometry_utils.py:199-228: pixels of image i with depths -> pixels and depths in image j."""
    hom = torch.cat([px, torch.ones_like(px[:, :1])], dim=-1) @ torch.linalg.inv(K_i).T * depth[:, None]
    X = hom @ T_ij[:3, :3].T + T_ij[:3, 3]
    uv = X @ K_j.T
    return uv[:, :2] / uv[:, 2:3].clamp(min=1e-6), X[:, 2]


def huber(diff, weights=None, delta=1.0):
    loss = torch.nn.functional.huber_loss(diff, torch.zeros_like(diff), reduction="none", delta=delta)
    if weights is not None:
        loss = loss * weights
    return loss.mean()


@contextlib.contextmanager
def injected_rng(jitter, grid, noises):
    """the renderer's torch.rand / torch.randn calls return the given draws, in call order"""
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
            return noises.pop(0).reshape(*size).to(kw.get("device", "cpu"))
        return real_randn(*size, **kw)

    torch.rand, torch.randn = rand, randn
    try:
        yield
    finally:
        torch.rand, torch.randn = real_rand, real_randn


class Workload:
    """graph, optimisers and `step(it)` (one training iteration; returns the loss tensor) of a config.
    `rays_last` counts the rays of an iteration that are rendered forward AND backward (the metric's
    "training rays"); the render_up_to_maxdepth rays of configs 3 / 4, which the reference renders
    forward-only under no_grad with 64 samples per network, are counted apart in `rays_fwd_only_last`."""

    def __init__(self, config, precision, device, rays=4096, optimizer="fused", batched=False, bucket_factory=None, seed=0, graph_capture=False):
        from sparf_amd.optim import FusedAdam
        self.graph_capture = bool(graph_capture)
        self.config, self.device, self.batched = config, device, batched
        s = SHAPES[config]
        self.B, self.H, self.W = s["B"], s["H"], s["W"]
        self.rays = rays
        # graph_capture: the whole step must be free of host -> device copies and host-side step state
        self.opt = opt = config_opt(config, precision, rays, **(dict(hip=dict(device_rng=True)) if graph_capture else {}))
        self.max_iter = opt.max_iter
        pose_gt, self.intr = cameras(config, device)
        self.image = analytic_images(pose_gt, self.intr, self.H, self.W, centre=(0.0, 0.0, 4.0) if s["layout"] == "forward" else (0.0, 0.0, 0.0))
        self.img_flat = self.image.flatten(2).permute(0, 2, 1).contiguous()            # [B,HW,3]
        torch.manual_seed(seed)
        if config == 1:
            self.graph = Graph(opt, device)
            pose_init = pose_gt
        else:
            g = torch.Generator().manual_seed(seed + 1)
            noise = (torch.randn(self.B, 6, generator=g) * 0.05).to(device)               # 'noisy_gt' initial poses (dtu/barf.py:31-32)
            pose_init = compose(se3_exp(noise), pose_gt)
            self.graph = PoseGraph(opt, device, pose_init)
        rng = s["rng"]
        self.data = edict(idx=torch.arange(self.B), image=self.image, intr=self.intr, pose=pose_gt,
                          depth_range=torch.tensor([list(rng)] * self.B, dtype=torch.float32, device=device))
        self.depth_min = float(rng[0])
        nets = [self.graph.nerf, self.graph.nerf_fine]
        if optimizer == "fused":                                                       # clip 0.1 + Adam (default_config.py:41-42, nerf_trainer.py:181-185)
            self.optim = FusedAdam(nets, lr=5e-4, max_grad_norm=0.1, device_step=self.graph_capture)
        else:
            self.optim = torch.optim.Adam([p for n in nets for p in n.parameters()], lr=5e-4)
        self.optim_pose = None
        if config != 1:                                                                                 # default_config.py:297
            try:
                self.optim_pose = torch.optim.Adam([self.graph.se3_refine], lr=1e-3, fused=True, capturable=self.graph_capture)        # one launch
            except (RuntimeError, TypeError):
                self.optim_pose = torch.optim.Adam([self.graph.se3_refine], lr=1e-3)
        self.optimizer = optimizer
        self.net_params = [p for net in nets for n, p in net.named_parameters() if n != "progress"]
        self.buckets = bucket_factory(self) if bucket_factory is not None else None
        self.rays_last = self.rays_fwd_only_last = 0
        # static synthetic correspondences for configs 3 / 4: pixel lists of a view pair, matched through a plane at depth 3
        if config in (3, 4):
            g = torch.Generator().manual_seed(seed + 2)
            n = rays // 2
            self.px_self = (torch.rand(n, 2, generator=g) * torch.tensor([self.W - 1.0, self.H - 1.0])).to(device)
            T = to44(pose_gt)
            self.T_gt_01 = T[1] @ torch.linalg.inv(T[0])
            with torch.no_grad():
                self.px_other, _ = project(self.px_self, torch.full((n,), 3.0, device=device), self.intr[0], self.intr[1], self.T_gt_01)
                self.px_other[:, 0].clamp_(0, self.W - 1)
                self.px_other[:, 1].clamp_(0, self.H - 1)
            self.conf = torch.rand(n, 1, generator=g).to(device)
            self.px_ref = (torch.rand(rays, 2, generator=g) * torch.tensor([self.W - 1.0, self.H - 1.0])).to(device)

    # ------------------------------------------------------------------ pieces
    def _photometric(self, ret, ray_idx):
        target = self.img_flat[:, ray_idx]
        if self.optimizer == "fused":       # MSE_loss on rgb + rgb_fine (base_losses.py:151-153, 303-311), one launch
            return ops.photometric_loss(ret.rgb, target, rgb_fine=ret.get("rgb_fine", None))
        loss = ((ret.rgb - target) ** 2).mean()
        if "rgb_fine" in ret:
            loss = loss + ((ret.rgb_fine - target) ** 2).mean()
        return loss

    def _sparf_requests(self, it, poses):
        """the four grad-mode render calls of a SPARF iteration that do not depend on each other"""
        opt, d, H, W = self.opt, self.data, self.H, self.W
        return [dict(pose=poses[0:1], intr=self.intr[0:1], pixels=self.px_self),                   # corres, self view
                dict(pose=poses[1:2], intr=self.intr[1:2], pixels=self.px_other),                  # corres, matching view
                dict(pose=poses[0:1].detach(), intr=self.intr[0:1], pixels=self.px_ref)]          # depth-consistency reference render (poses detached, depth_cons_loss.py:176)

    def _sparf_losses(self, it, poses, rets):
        """correspondence (corres_loss.py:150-200) and depth-consistency (depth_cons_loss.py:150-300)
        terms from the renders of _sparf_requests, issuing the two dependent renders."""
        opt, d, H, W, g = self.opt, self.data, self.H, self.W, self.graph
        ret_self, ret_other, ret_ref = rets
        key = "depth_fine" if "depth_fine" in ret_self else "depth"
        d_self, d_other = ret_self[key].reshape(-1), ret_other[key].reshape(-1)
        T = to44(poses)
        T_01 = T[1] @ torch.linalg.inv(T[0])
        uv, _ = project(self.px_self, d_self, self.intr[0], self.intr[1], T_01)
        uv2, _ = project(self.px_other, d_other, self.intr[1], self.intr[0], torch.linalg.inv(T_01))
        loss_corres = huber(uv - self.px_other, self.conf) + huber(uv2 - self.px_self, self.conf)
        # depth consistency: back-project the reference render, look at it from an unseen pose
        if ret_ref is None:
            # (separate calls: the depth-consistency module issues its reference render AFTER the correspondence module has read its two
            # renders -- depth_cons_loss.py:192 vs corres_loss.py:158-166 -- so only those two can meet in one lazily batched launch set)
            q = self._sparf_requests(it, poses)[2]
            ret_ref = g.render_image_at_specific_pose_and_rays(opt, d, q["pose"][0], q["intr"][0], H, W, pixels=q["pixels"], mode="train", iter=it)
        depth_ref = ret_ref[key].reshape(-1)
        Tn = T.detach()
        c2w_ref = torch.linalg.inv(Tn[0])
        hom = torch.cat([self.px_ref, torch.ones_like(self.px_ref[:, :1])], dim=-1) @ torch.linalg.inv(self.intr[0]).T * depth_ref[:, None]
        pts_w = hom @ c2w_ref[:3, :3].T + c2w_ref[:3, 3]
        if getattr(self, "_shift", None) is None:       # the (fixed) offset of the unseen pose: a constant of the workload
            self._shift = se3_exp(torch.tensor([[0.0, 0.06, 0.0, 0.15, 0.0, 0.0]], device=self.device))
        shift = self._shift
        pose_unseen = compose(shift, Tn[0:1, :3])[0]
        X = pts_w @ pose_unseen[:, :3].T + pose_unseen[:, 3]
        uvw = X @ self.intr[0].T
        pts_img, depth_pseudo = uvw[:, :2] / uvw[:, 2:3].clamp(min=1e-6), X[:, 2]
        valid = (pts_img[:, 0] >= 0) & (pts_img[:, 1] >= 0) & (pts_img[:, 0] <= W - 1) & (pts_img[:, 1] <= H - 1) & (depth_pseudo >= self.depth_min)
        pts_img, depth_pseudo = pts_img[valid], depth_pseudo[valid]
        n_max = pts_img.shape[0]
        with torch.no_grad():
            rm = g.render_up_to_maxdepth_at_specific_pose_and_rays(opt, d, pose_unseen, self.intr[0], H, W, depth_max=depth_pseudo.detach(),
                                                                   pixels=pts_img.detach(), mode="train", iter=it)
            vis = (rm["all_cumulated_fine"] if "all_cumulated_fine" in rm else rm["all_cumulated"]).reshape(-1, 1)
        keep = vis.reshape(-1) >= 0.2
        pts_img, depth_pseudo, vis = pts_img[keep], depth_pseudo[keep], vis[keep]
        n_last = pts_img.shape[0]
        loss_dc = torch.zeros((), device=self.device)
        if n_last > 0:
            # (the pixel list keeps its gradient, as in depth_cons_loss.py:291: the projections depend on the reference render's depth
            # and the reference's ray generation is differentiable in them -- found by running the reference's own loss on this
            # renderer, tests/test_reference_callers_gpu.py)
            rs = g.render_image_at_specific_pose_and_rays(opt, d, pose_unseen, self.intr[0], H, W, pixels=pts_img, mode="train", iter=it)
            acc = rs.opacity.reshape(-1, 1).detach()
            loss_dc = huber(depth_pseudo.reshape(-1) - rs.depth.reshape(-1), (vis * acc).reshape(-1))
        return loss_corres, loss_dc, n_max, n_last

    # ------------------------------------------------------------------ the step as one hipGraph
    def capture(self, warmup=3):
        """Capture one training iteration (ray generation -> both passes -> loss -> backward -> clip + Adam, ~40 launches)
        in a hipGraph and return `replay()`: the ray selection of the next step is drawn outside the graph into a static
        index buffer, everything else -- stratified jitter, the fine grid, density noise (device RNG, philox offsets
        advanced per replay by torch's graph-safe generator), the optimiser's step count (sparf_adam_step_dev) -- lives
        on the device.  Configs 1 / 2 only: the SPARF call mix of configs 3 / 4 has data-dependent ray counts.
        With a gradient exchange (data parallelism, `bucket_factory`) the step is TWO graphs -- forward + backward, then clip +
        Adam -- with the one all-reduce of the step issued eagerly between them on the captured gradient buffers (round 4
        refused to capture when a bucket existed, so the strong-scaling leg of `bench.py --gpus 8` ran its 512-ray steps eagerly,
        at the host-bound rate: VERDICT r04 weak-7)."""
        if self.config not in (1, 2) or not self.graph_capture or self.optimizer != "fused":
            raise RuntimeError("Workload.capture: configs 1 / 2, graph_capture=True, fused optimiser")
        R = self.rays // self.B
        self._ray_idx = torch.empty(R, dtype=torch.int64, device=self.device)
        self._loss_static = None

        def fwd_bwd():
            self._loss_static = self._step_with(self._ray_idx, update=False)

        def update():
            self.optim.step()
            if self.optim_pose is not None:
                self.optim_pose.step()

        def draw():
            self._ray_idx.copy_(torch.randperm(self.H * self.W, device=self.device)[:R])

        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):                  # allocator warm-up + optimiser state creation outside the capture
                draw()
                fwd_bwd()
                if self.buckets is not None:
                    self.buckets(self._loss_static)
                update()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        self._loss_static = None                                   # the warm-up ran on the side stream: no autograd state of it survives into the capture
        for net in (self.graph.nerf, self.graph.nerf_fine):
            net.release_autograd_cache()
        draw()
        g1, g2 = torch.cuda.CUDAGraph(), None
        # (with a process group alive its watchdog thread queries events while we capture: thread-local capture mode keeps other threads'
        # runtime calls legal)
        mode = dict(capture_error_mode="thread_local") if self.buckets is not None else {}
        if self.buckets is None:
            with torch.cuda.graph(g1):
                fwd_bwd()
                update()
        else:
            with torch.cuda.graph(g1, **mode):
                fwd_bwd()
            # (the gradients the update reads are the tensors the first capture left in p.grad: static memory of its pool, rewritten
            # by every replay and reduced in place by the exchange in between)
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, **mode):
                update()

        def replay():
            draw()
            g1.replay()
            if g2 is not None:
                self.buckets(self._loss_static)
                g2.replay()
            self.rays_last = self.B * R
            return self._loss_static
        self._graph, self._graph_update = g1, g2
        return replay

    def _step_with(self, ray_idx, it=100000, update=True):
        """config 1 / 2 iteration on given pixel indices (shared by step() and the captured graph); update=False: forward +
        backward only (the captured step of a data-parallel run exchanges gradients before the optimiser)"""
        g, opt, d = self.graph, self.opt, self.data
        self.optim.zero_grad(set_to_none=True)
        if self.optim_pose is not None:
            self.optim_pose.zero_grad(set_to_none=True)
        if opt.barf_c2f is not None:
            g.nerf.progress.data.fill_(it / self.max_iter)
            g.nerf_fine.progress.data.fill_(it / self.max_iter)
        pose = g.get_w2c_pose(opt, d, mode="train") if self.config != 1 else d.pose
        ret = g.render(opt, pose, H=self.H, W=self.W, intr=self.intr, ray_idx=ray_idx, depth_range=g._depth_range(opt, d), iter=it, mode="train")
        loss = self._photometric(ret, ray_idx)
        loss.backward()
        if update:
            self.optim.step()
            if self.optim_pose is not None:
                self.optim_pose.step()
        return loss

    # ------------------------------------------------------------------ one iteration
    def step(self, it=100000):
        """it: training iteration the step pretends to be at (past every start gate; c2f progress 0.5)"""
        g, opt, d = self.graph, self.opt, self.data
        self.optim.zero_grad(set_to_none=True)
        if self.optim_pose is not None:
            self.optim_pose.zero_grad(set_to_none=True)
        if opt.barf_c2f is not None:                     # nerf_trainer.py:273-275
            g.nerf.progress.data.fill_(it / self.max_iter)
            g.nerf_fine.progress.data.fill_(it / self.max_iter)
        nrays = 0
        if self.config == 1:
            R = self.rays // self.B
            ray_idx = torch.randperm(self.H * self.W, device=self.device)[:R]
            ret = g.render(opt, d.pose, H=self.H, W=self.W, intr=self.intr, ray_idx=ray_idx, depth_range=d.depth_range[0], iter=it, mode="train")
            loss = self._photometric(ret, ray_idx)
            nrays = self.B * R
        elif self.config == 2:
            ret = g.forward(opt, d, iter=it, mode="train")                      # randperm rays, poses from get_w2c_pose
            loss = self._photometric(ret, ret.ray_idx)
            nrays = self.B * ret.ray_idx.numel()
        else:
            poses = g.get_w2c_pose(opt, d, mode="train")
            H, W = self.H, self.W
            R = self.rays // self.B
            ray_idx = torch.randperm(H * W, device=self.device)[:R]
            rng = g._depth_range(opt, d)
            reqs = self._sparf_requests(it, poses)
            if self.batched:
                allreq = [dict(pose=poses, H=H, W=W, intr=self.intr, ray_idx=ray_idx, depth_range=rng, mode="train")] + \
                         [dict(q, H=H, W=W, depth_range=rng, mode="train") for q in reqs]
                rets = g.render_batch(opt, allreq, iter=it)
                ret, rets = rets[0], rets[1:]
            else:
                ret = g.render(opt, poses, H=H, W=W, intr=self.intr, ray_idx=ray_idx, depth_range=rng, iter=it, mode="train")
                rets = [g.render_image_at_specific_pose_and_rays(opt, d, q["pose"][0], q["intr"][0], H, W, pixels=q["pixels"], mode="train", iter=it)
                        for q in reqs[:2]] + [None]          # (the third is issued where the depth-consistency loss issues it: _sparf_losses)
            loss_c, loss_d, n_max, n_last = self._sparf_losses(it, poses, rets)
            loss = self._photometric(ret, ray_idx) + 1e-3 * loss_c + 1e-3 * loss_d      # loss_weight.corres = depth_cons = -3 (10^)
            nrays = self.B * R + sum(q["pixels"].shape[0] for q in reqs) + n_last
            self.rays_fwd_only_last = n_max
        loss.backward()
        if self.buckets is not None:
            self.buckets(loss)
        if self.optimizer == "fused":
            self.optim.step()
        else:
            for net in (g.nerf, g.nerf_fine):
                torch.nn.utils.clip_grad_norm_(net.parameters(), 0.1)
            self.optim.step()
        if self.optim_pose is not None:
            self.optim_pose.step()
        self.rays_last = nrays
        return loss
