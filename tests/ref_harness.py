"""Harness that runs the REFERENCE's own, unmodified training-iteration code -- `RaySamplingStrategy`
(sampling_strategies.py:23-188), `define_loss` -> `BasePhotoandReguLoss` (base_losses.py:243-323),
`CorrespondencesPairRenderDepthAndGet3DPtsAndReproject` (corres_loss.py:27-223 on base_corres_loss.py:30-375),
`DepthConsistencyLoss` (depth_cons_loss.py:31-321), the `class Graph(Graph)` body of
joint_pose_nerf_trainer.py:710-749 and the reference pose network -- on top of a renderer `Graph` handed
to it: the reference's `source.models.renderer.Graph`, or `sparf_amd.renderer.Graph`.

TEST INFRASTRUCTURE (VERDICT r03 missing-1 / next-1).  The reference tree is imported from $SPARF_REFERENCE_ROOT (a checkout
or an archive made by the opt-in tool `oracle/stage_reference.py`) or from the build container's /root/reference.
Five modules the reference imports but this image lacks are stubbed -- none of them is on the path the
losses execute: `lpips` (a VGG metric constructed at import, base_losses.py:139), `cv2`, `imageio`
(dataset / visualisation helpers), `third_party.DenseMatching.utils_flow.pixel_wise_mapping`
(a plotting helper of correspondence_utils.py), `easydict` (compat/, the repo's own dict class).

What stands in for data that is not here:
  * the scene: `bench_workloads.cameras / analytic_images` (a textured sphere in front of a gradient);
  * PDC-Net ("matches precomputed", BASELINE configs 3 / 4): `SphereFlowNet`, an object with the three
    methods `CorrespondenceBasedLoss.compute_correspondences` calls (base_corres_loss.py:64-140) that
    returns the scene's exact correspondence maps + a confidence map -- the fields
    base_corres_loss.py:130-147 stores.  It feeds INPUTS to the loss; no loss arithmetic is restated.

Both runs of a comparison see identical random draws: `DrawTape` records every torch.rand / randn /
randn_like / randperm and np.random.rand / randint result of the first run and replays it, FIFO per
(kind, size), in the second -- the two renderers draw the same tensors in the same order
(renderer.py:405-407, :439, frequency_nerf.py:191-192).  `DrawTape(seed=s)` draws the first run's numbers from
`np.random.RandomState(s)` -- including the `np.random` calls of the loss modules (depth_cons_loss.py:57,181,
base_corres_loss.py:164), which no torch seed reaches: the recorded run itself is then the same on every box
(round 4 seeded torch only, and the reference's data-dependent ray counts wandered from lease to lease).
"""
import contextlib
import importlib
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "compat")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle.stage_reference import import_root, staged_root  # noqa: E402


def reference_root():
    """the staged archive / reference tree if there is one (cheap test for skip marks); install_reference() makes it importable"""
    return staged_root()


def install_reference():
    """Put the reference tree on sys.path and stub the modules this image lacks.  -> root or None"""
    root = import_root()
    if root is None:
        return None
    if root not in sys.path:
        sys.path.append(root)              # behind the repo: `oracle`, `tests`, `sparf_amd` keep resolving here

    def stub(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        try:
            return importlib.import_module(name)
        except Exception:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            m.__sparf_stub__ = True
            sys.modules[name] = m
            return m

    class _NoLPIPS:                                        # base_losses.py:139 builds one at import time; never called by a loss
        def __init__(self, *a, **k):
            pass

        def to(self, device):
            return self

    stub("lpips", LPIPS=_NoLPIPS)
    stub("cv2")
    stub("imageio")
    tp = stub("third_party")
    if not hasattr(tp, "__path__"):
        tp.__path__ = []
    if True:       # DenseMatching is an un-vendored (empty) submodule in the reference checkout and absent from the staged archive
        dm = stub("third_party.DenseMatching")
        dm.__path__ = getattr(dm, "__path__", [])
        uf = stub("third_party.DenseMatching.utils_flow")
        uf.__path__ = getattr(uf, "__path__", [])
        stub("third_party.DenseMatching.utils_flow.pixel_wise_mapping", warp_with_mapping=lambda *a, **k: (_ for _ in ()).throw(
            RuntimeError("warp_with_mapping is a plotting helper; not available in the test harness")))
    return root


# ---------------------------------------------------------------------------------------------- random draws
class DrawTape:
    """record / replay of every random draw the training iteration makes.
    seed: draw the RECORDED run's numbers from np.random.RandomState(seed) (Mersenne twister: the same stream on every box and
    numpy version) instead of torch's / numpy's global generators; None: the process's generators as they stand."""

    def __init__(self, seed=None):
        self.fifo = {}
        self.mode = None
        self.log = []            # keys of the recorded draws in order
        self.values = []         # the recorded values, parallel to `log` (tests/callers_tape.py slices them per render call)
        self.resized = []
        self.seed = seed
        self.rs = np.random.RandomState(seed) if seed is not None else None

    # the seeded generators: float32 like torch's, uniform draws kept inside [0, 1) after the cast
    def np_uniform(self, shape):
        x = self.rs.random_sample(tuple(shape)).astype(np.float32)
        return torch.from_numpy(np.minimum(x, np.float32(1.0) - np.float32(2.0 ** -24)))

    def np_normal(self, shape):
        return torch.from_numpy(self.rs.standard_normal(tuple(shape)).astype(np.float32))

    def np_perm(self, n):
        return torch.from_numpy(self.rs.permutation(int(n)).astype(np.int64))

    def _take(self, key, make, like_device=None):
        if self.mode == "record":
            v = make()
            kept = v.detach().cpu().clone() if torch.is_tensor(v) else v
            self.fifo.setdefault(key, []).append(kept)
            self.log.append(key)
            self.values.append(kept)
            if torch.is_tensor(v) and like_device is not None and self.rs is not None:
                v = v.to(like_device)
            return v
        q = self.fifo.get(key)
        if not q and key[0] in ("rand", "randn"):
            # A data-dependent ray count differs between the two runs (a threshold of depth_cons_loss.py:254-274 decided
            # differently for a ray whose value sits within the renderers' distance of it): replay the recorded draw of
            # the nearest size, cut or topped up with fresh numbers, and say so -- the comparison of that call is then
            # statistical, not elementwise (tests/test_reference_callers_gpu.py treats it apart).
            near = [k for k, lst in self.fifo.items() if lst and k[0] == key[0] and abs(k[1] - key[1]) <= 0.02 * key[1]]
            if near:
                k2 = min(near, key=lambda k: abs(k[1] - key[1]))
                rec = self.fifo[k2].pop(0).reshape(-1)
                v = make().detach().cpu().reshape(-1).clone()
                n = min(v.numel(), rec.numel())
                v[:n] = rec[:n]
                self.resized.append((k2, key))
                return v.to(like_device) if like_device is not None else v
        if not q:
            raise AssertionError(f"DrawTape: the replayed run asks for a draw {key} the recorded run never made "
                                 f"(recorded: {sorted(set(self.log))})")
        v = q.pop(0)
        if torch.is_tensor(v) and like_device is not None:
            v = v.to(like_device)
        return v

    @contextlib.contextmanager
    def run(self, mode):
        assert mode in ("record", "replay")
        self.mode = mode
        real = dict(rand=torch.rand, randn=torch.randn, randn_like=torch.randn_like, randperm=torch.randperm,
                    np_rand=np.random.rand, np_randint=np.random.randint)
        tape = self
        seeded = self.rs is not None

        def numel(size):
            if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
                size = tuple(size[0])
            n = 1
            for s in size:
                n *= int(s)
            return n, tuple(int(s) for s in size)

        def rand(*size, **kw):
            n, shape = numel(size)
            dev = kw.get("device", "cpu")
            return tape._take(("rand", n), (lambda: tape.np_uniform(shape)) if seeded else (lambda: real["rand"](*size, **kw)), dev).reshape(shape)

        def randn(*size, **kw):
            n, shape = numel(size)
            dev = kw.get("device", "cpu")
            return tape._take(("randn", n), (lambda: tape.np_normal(shape)) if seeded else (lambda: real["randn"](*size, **kw)), dev).reshape(shape)

        def randn_like(t, **kw):
            return tape._take(("randn", t.numel()), (lambda: tape.np_normal(tuple(t.shape))) if seeded else (lambda: real["randn_like"](t, **kw)),
                              t.device).reshape(t.shape)

        def randperm(n, **kw):
            return tape._take(("randperm", int(n)), (lambda: tape.np_perm(n)) if seeded else (lambda: real["randperm"](n, **kw)), kw.get("device", "cpu"))

        def np_rand(*a):
            return tape._take(("np.rand", a), (lambda: (float(tape.rs.random_sample()) if not a else tape.rs.random_sample(a))) if seeded
                              else (lambda: real["np_rand"](*a)))

        def np_randint(*a, **k):
            return tape._take(("np.randint", a), (lambda: tape.rs.randint(*a, **k)) if seeded else (lambda: real["np_randint"](*a, **k)))

        torch.rand, torch.randn, torch.randn_like, torch.randperm = rand, randn, randn_like, randperm
        np.random.rand, np.random.randint = np_rand, np_randint
        try:
            yield self
        finally:
            torch.rand, torch.randn, torch.randn_like, torch.randperm = real["rand"], real["randn"], real["randn_like"], real["randperm"]
            np.random.rand, np.random.randint = real["np_rand"], real["np_randint"]
            self.mode = None

    def leftover(self):
        return {k: len(v) for k, v in self.fifo.items() if v}


# ---------------------------------------------------------------------------------------------- the scene
SETTINGS = {
    # name -> (settings module, bench_workloads shape id, what BASELINE config it is)
    "dtu_nerf": ("train_settings.nerf_training_w_gt_poses.dtu.nerf", 1, "BASELINE configs 0 / 1: fixed GT poses, plain Graph (nerf_trainer.py:112-114)"),
    "dtu_barf": ("train_settings.joint_pose_nerf_training.dtu.barf", 2, "BASELINE config 2"),
    "llff_sparf": ("train_settings.joint_pose_nerf_training.llff.sparf", 3, "BASELINE config 3"),
    "replica_sparf": ("train_settings.joint_pose_nerf_training.replica.sparf", 4, "BASELINE config 4"),
}


def load_settings(name, rays=None, samples=(64, 128), scene_hw=None):
    """The reference's own `get_config()` of a BASELINE config; BASELINE's sample counts (64 coarse + 128 fine)
    and ray batch on top (the settings files ship 128 / no fine network for LLFF, 1024-2048 rays)."""
    install_reference()
    opt = importlib.import_module(SETTINGS[name][0]).get_config()
    opt.nerf.sample_intvs, opt.nerf.sample_intvs_fine = samples
    opt.nerf.fine_sampling = True
    if rays is not None:
        opt.nerf.rand_rays = rays
    opt.device = "cuda" if torch.cuda.is_available() else "cpu"
    if scene_hw is not None:
        opt._scene_hw = tuple(scene_hw)            # the harness's scene at a reduced image size (CPU self-test)
    return opt


class SphereFlowNet:
    """Stands in for PDC-Net: exact correspondences of the analytic scene (a unit sphere at `centre`) between
    every ordered view pair, confidence 1 where the surface point is seen by both views, 0 elsewhere."""

    def __init__(self, pose_w2c, intr, H, W, centre):
        self.pose, self.intr, self.H, self.W = pose_w2c, intr, H, W
        self.centre = torch.tensor(centre, dtype=torch.float32, device=pose_w2c.device)
        B = pose_w2c.shape[0]
        pairs = [(i, j) for i in range(B) for j in range(B) if i != j]
        self.combi_list = torch.tensor(pairs, dtype=torch.long).T            # [2, N]: row 0 target (= "self"), row 1 source

    def _maps(self, combi):
        dev = self.pose.device
        H, W = self.H, self.W
        ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
        pix = torch.stack([xs, ys, torch.ones_like(xs)], dim=-1).reshape(-1, 3)
        corres, conf = [], []
        for k in range(combi.shape[1]):
            i, j = int(combi[0, k]), int(combi[1, k])
            Ri, ti = self.pose[i, :, :3], self.pose[i, :, 3]
            Rj, tj = self.pose[j, :, :3], self.pose[j, :, 3]
            o = -(Ri.T @ ti) - self.centre
            d = (pix @ torch.linalg.inv(self.intr[i]).T) @ Ri
            d = d / d.norm(dim=-1, keepdim=True)
            bq = d @ o
            disc = bq * bq - (o @ o - 1.0)
            tt = -bq - disc.clamp(min=0).sqrt()
            p = o + d * tt[:, None]                               # surface point relative to the sphere centre = its normal
            cam_j = -(Rj.T @ tj) - self.centre
            seen = (disc > 0) & (tt > 0) & (((cam_j - p) * p).sum(-1) > 0.05)
            X = (p + self.centre) @ Rj.T + tj
            uv = X @ self.intr[j].T
            uv = uv[:, :2] / uv[:, 2:3].clamp(min=1e-6)
            ok = seen & (uv[:, 0] >= 0) & (uv[:, 0] <= W - 1) & (uv[:, 1] >= 0) & (uv[:, 1] <= H - 1)
            uv = torch.where(ok[:, None], uv, torch.full_like(uv, -10.0))
            corres.append(uv.reshape(H, W, 2).permute(2, 0, 1))
            conf.append(ok.float().reshape(1, H, W))
        return torch.stack(corres), torch.stack(conf)

    def compute_flow_and_confidence_map_of_combi_list(self, images, combi_list_tar_src, plot=False, use_homography=False):
        corres, conf = self._maps(combi_list_tar_src)
        return corres, conf, None

    def compute_flow_and_confidence_map_and_cc_of_combi_list(self, images, combi_list_tar_src, plot=False, use_homography=False):
        corres, conf = self._maps(combi_list_tar_src)
        return corres, conf, conf.clone(), None

    def visualize_mapping_combinations(self, images, mapping_est, batched_conf_map, combi_list, save_path=None):
        return np.zeros((8, 8, 3), dtype=np.uint8)


def make_scene(name, opt, device, seed=0):
    """cameras, images, initial (noisy) poses and the reference's `train_data` / `data_dict` records for a settings name"""
    import bench_workloads as BW
    from easydict import EasyDict as edict
    cfg = SETTINGS[name][1]
    s = BW.SHAPES[cfg]
    H, W, B = s["H"], s["W"], s["B"]
    if opt.get("_scene_hw"):
        H, W = opt["_scene_hw"]
    pose_gt, intr = BW.cameras(cfg, device)
    if (H, W) != (s["H"], s["W"]):
        intr = intr.clone()
        intr[:, 0] *= W / s["W"]
        intr[:, 1] *= H / s["H"]
    centre = (0.0, 0.0, 4.0) if s["layout"] == "forward" else (0.0, 0.0, 0.0)
    image = BW.analytic_images(pose_gt, intr, H, W, centre=centre)
    g = torch.Generator().manual_seed(seed + 1)
    noise = (torch.randn(B, 6, generator=g) * 0.03).to(device)
    pose_init = BW.compose(BW.se3_exp(noise), pose_gt)
    rng = s["rng"]
    depth_range = torch.tensor([list(rng)] * B, dtype=torch.float32, device=device)
    allv = edict(idx=torch.arange(B, device=device), image=image, intr=intr, pose=pose_gt, depth_range=depth_range)
    train_data = edict(all=allv)
    flow_net = SphereFlowNet(pose_gt, intr, H, W, centre)
    return edict(B=B, H=H, W=W, pose_gt=pose_gt, pose_init=pose_init, intr=intr, image=image, train_data=train_data, flow_net=flow_net,
                 depth_range=depth_range)


# ---------------------------------------------------------------------------------------------- the graph under test
def joint_graph_class(base_graph_cls):
    """`class Graph(Graph)` of joint_pose_nerf_trainer.py:710-749, its body taken from the reference file at run time
    (never copied into this repo), on top of the given base class"""
    install_reference()
    import source.utils.camera as camera
    from source.utils.geometry.align_trajectories import (backtrack_from_aligning_and_scaling_to_first_cam,
                                                          backtrack_from_aligning_the_trajectory)
    from typing import Any, Dict
    from oracle.stage_reference import read_text
    src = read_text("source/training/joint_pose_nerf_trainer.py")
    body = src[src.index("class Graph(Graph):"):]
    ns = dict(Graph=base_graph_cls, camera=camera, torch=torch, Dict=Dict, Any=Any,
              backtrack_from_aligning_and_scaling_to_first_cam=backtrack_from_aligning_and_scaling_to_first_cam,
              backtrack_from_aligning_the_trajectory=backtrack_from_aligning_the_trajectory)
    exec(body, ns)
    return ns["Graph"]


def build_graph(kind, opt, scene, device, state=None, precision=None):
    """kind: 'reference' -> source.models.renderer.Graph, 'hip' -> sparf_amd.renderer.Graph; both wrapped in the reference's
    joint-pose subclass with the reference's pose network (camera.pose_parametrization, default 'two_columns')."""
    install_reference()
    if kind == "reference":
        from source.models.renderer import Graph as Base
    else:
        from sparf_amd.renderer import Graph as Base
        if precision is not None:
            from easydict import EasyDict as edict
            opt = edict(opt)
            opt.hip = edict(precision=precision)
    if opt.model == "nerf_gt_poses":                    # NerfTrainerPerScene.build_nerf_net: the renderer itself, poses from the data
        graph = Base(opt, device)
    else:
        from source.models.poses_models.two_columns import FirstTwoColunmnsPoseParameters
        Sub = joint_graph_class(Base)
        pose_net = FirstTwoColunmnsPoseParameters(opt, nbr_poses=scene.B, initial_poses_w2c=scene.pose_init, device=torch.device(device))
        graph = Sub(opt, device, pose_net)
    if state is not None:
        graph.load_state_dict(state, strict=True)
    graph.train()
    return graph, opt


class CallLog:
    """records every render call the loss modules make on the graph (method, pixel / ray counts, outputs)"""

    METHODS = ("render", "render_to_max")

    def __init__(self, graph):
        self.calls = []
        for name in self.METHODS:
            fn = getattr(graph, name)

            def wrapped(*a, __fn=fn, __name=name, **k):
                ret = __fn(*a, **k)
                px, ix = k.get("pixels"), k.get("ray_idx")
                n = (px.shape[-2] if px is not None else ix.shape[-1] if ix is not None else None)
                keep = {key: ret[key].detach().float().cpu() for key in ("rgb", "depth", "opacity", "rgb_fine", "depth_fine", "opacity_fine",
                                                                         "all_cumulated", "all_cumulated_fine") if key in ret}
                self.calls.append(dict(method=__name, n=n, grad=torch.is_grad_enabled(), out=keep))
                return ret

            setattr(graph, name, wrapped)


def training_iteration(graph, opt, scene, iteration, tape, mode, per_term_grads=False):
    """One iteration as nerf_trainer.py:224-245 runs it: sample rays -> render -> poses into data_dict -> loss_module.compute_loss
    -> backward.  -> (loss dict of floats, gradient dict, call log)"""
    install_reference()
    from easydict import EasyDict as edict
    from source.training.core.loss_factory import define_loss
    from source.training.core.sampling_strategies import RaySamplingStrategy
    dev = scene.image.device
    log = CallLog(graph)
    with tape.run(mode):
        loss_module = define_loss(opt.loss_type, opt, graph, scene.train_data, dev, flow_net=scene.flow_net)
        sampler = RaySamplingStrategy(opt, data_dict=scene.train_data.all, device=dev)
        data_dict = edict(scene.train_data.all)
        data_dict.iter = iteration
        if opt.barf_c2f is not None and opt.apply_cf_pe:                           # nerf_trainer.py:271-275
            graph.nerf.progress.data.fill_(iteration / opt.max_iter)
            graph.nerf_fine.progress.data.fill_(iteration / opt.max_iter)
        rays = sampler(opt.nerf.rand_rays, sample_in_center=iteration < opt.precrop_iters)
        output_dict = graph.render_image_at_specific_rays(opt, data_dict, ray_idx=rays, iter=iteration, mode="train")
        data_dict.poses_w2c = graph.get_w2c_pose(opt, data_dict, mode="train")
        loss_dict, stats, _ = loss_module.compute_loss(opt, data_dict, output_dict, mode="train", plot=False, iteration=iteration)
        for p in graph.parameters():
            p.grad = None
        per_term = {}
        if per_term_grads:          # the gradient of every loss term on its own (diagnosis: which caller's backward differs)
            named = [(n, p) for n, p in graph.named_parameters() if p.requires_grad]
            for k in ("render", "corres", "depth_cons"):
                if k in loss_dict and loss_dict[k].requires_grad:
                    gs = torch.autograd.grad(loss_dict[k], [p for _, p in named], retain_graph=True, allow_unused=True)
                    per_term[k] = {n: g.detach().float().cpu().clone() for (n, _), g in zip(named, gs) if g is not None}
        loss_dict["all"].backward()
    losses = {k: float(v.detach()) for k, v in loss_dict.items() if torch.is_tensor(v) and v.dim() == 0}
    grads = {n: p.grad.detach().float().cpu().clone() for n, p in graph.named_parameters() if p.grad is not None}
    if per_term_grads:
        return losses, grads, log.calls, per_term
    return losses, grads, log.calls


def rel_max(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def compare(ref, test):
    """ref / test = results of training_iteration -> dict of error numbers"""
    (l0, g0, c0), (l1, g1, c1) = ref[:3], test[:3]
    out = dict(loss={k: dict(ref=l0[k], test=l1.get(k), rel=abs(l1.get(k, float("nan")) - l0[k]) / (abs(l0[k]) + 1e-12)) for k in l0})
    out["calls"] = dict(ref=[(c["method"], c["n"], c["grad"]) for c in c0], test=[(c["method"], c["n"], c["grad"]) for c in c1])
    per_call = []
    for a, b in zip(c0, c1):
        if a["n"] != b["n"] or a["method"] != b["method"]:
            per_call.append(dict(mismatch=(a["method"], a["n"], b["method"], b["n"])))
            continue
        per_call.append({k: rel_max(b["out"][k], a["out"][k]) for k in a["out"] if k in b["out"]})
    out["per_call"] = per_call
    net = [n for n in g0 if not n.startswith("pose_net.")]
    pose = [n for n in g0 if n.startswith("pose_net.")]
    out["grad_worst_tensor"] = max(rel_l2(g1[n], g0[n]) for n in net)
    out["grad_worst_name"] = max(net, key=lambda n: rel_l2(g1[n], g0[n]))
    out["grad_all"] = rel_l2(torch.cat([g1[n].reshape(-1) for n in net]), torch.cat([g0[n].reshape(-1) for n in net]))
    out["grad_pose"] = max((rel_max(g1[n], g0[n]) for n in pose), default=None)        # None: fixed poses (no pose network)
    out["missing_grads"] = sorted(set(g0) - set(g1))
    out["grad_per_tensor"] = {n: rel_l2(g1[n], g0[n]) for n in g0 if n in g1}
    if len(ref) > 3 and len(test) > 3:
        out["per_term"] = {k: dict(worst=max(rel_l2(test[3][k][n], g) for n, g in ref[3][k].items() if n in test[3][k]),
                                   all=rel_l2(torch.cat([test[3][k][n].reshape(-1) for n in ref[3][k] if n in test[3][k]]),
                                              torch.cat([g.reshape(-1) for n, g in ref[3][k].items() if n in test[3][k]])),
                                   norm=float(torch.cat([g.reshape(-1) for g in ref[3][k].values()]).norm()))
                           for k in ref[3] if k in test[3]}
    return out
