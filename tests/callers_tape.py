"""TEACHER-FORCED comparison of a renderer with the reference, call by call.

One training iteration of the reference's OWN loss modules (tests/ref_harness.py: `RaySamplingStrategy`, `define_loss` ->
photometric / correspondence / depth-consistency losses, the joint-pose `class Graph(Graph)`) is run on top of the REFERENCE
`Graph`, and everything the loss code asked of the renderer and did with the answer is taped:

    per render call   the arguments (pose, intrinsics, pixels | ray_idx, depth range | per-ray depth_max, iter, mode, grad mode),
                      the random draws it consumed (jitter, fine grid, density noise), the outputs the callers read
                      (rgb / depth / opacity (+ _fine), all_cumulated(_fine) of render_to_max), the UPSTREAM GRADIENT the loss
                      sent back into each of those outputs, and the gradient that came out of the call at its pose and pixel inputs
    per iteration     the loss terms, the gradients of both networks

`replay` then drives another renderer (the HIP `Graph`) with the TAPED arguments, call by call: every call is compared on
identical inputs -- also the calls whose pixel lists / depth caps the free-running chain derives from an earlier render's output
(depth_cons_loss.py:199-201, 254-262, :267, :291), where round 4 compared two different inputs -- and the taped upstream gradients
are pushed back through it, so the parameter gradients it accumulates over the calls are the iteration's gradients
(d loss / d theta = sum over calls of J_call^T g_call: the hooks tape the TOTAL gradient of every output, the path through the
pixel coordinates of a later call included).

A tape serialises to a small .npz (tests/golden/callers_tape_*.npz, made from the reference on the CPU by
tests/golden/make_callers_tape.py): the draws are regenerated from the tape's numpy seed, the weights from `seeded_state`, the
big gradient tensors are kept as a fixed random subset of their entries + their norms.  The replay on the GPU box then needs no
reference code at all.
"""
import contextlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "compat")):
    if p not in sys.path:
        sys.path.insert(0, p)

OUT_KEYS = ("rgb", "depth", "opacity", "rgb_fine", "depth_fine", "opacity_fine", "all_cumulated", "all_cumulated_fine")
SUBSET = 8192                  # entries kept of a gradient tensor with more than FULL_BELOW elements
FULL_BELOW = 40000
LAYER_SHAPES = [(256, 63), (256, 256), (256, 256), (256, 256), (256, 319), (256, 256), (256, 256), (257, 256), (128, 283), (3, 128)]


# ---------------------------------------------------------------------------------------------- weights
def seeded_state(seed, like=None):
    """state_dict of one NeRF (mlp_feat.0-7, mlp_rgb.0-1; `progress` left to the caller) drawn from np.random.RandomState(seed):
    Xavier-uniform at the reference initialisation's scale (frequency_nerf.py:136-147: relu gain except the density row and the
    colour output) with small non-zero biases (the reference starts them at 0: a bias gradient path worth exercising) -- the same
    numbers on every box, so a committed tape needs no 2 MB of weights next to it."""
    rs = np.random.RandomState(seed)
    gain = float(np.sqrt(2.0))
    sd = {}
    for i, (o, k) in enumerate(LAYER_SHAPES):
        name = f"mlp_feat.{i}" if i < 8 else f"mlp_rgb.{i - 8}"
        g = np.full((o, 1), gain)
        if i == 7:
            g[0, 0] = 1.0                      # the raw-density row: xavier_uniform_(weight[:1]) without gain
        if i == 9:
            g[:] = 1.0
        a = g * np.sqrt(6.0 / (o + k))
        if i == 7:
            a[0, 0] = np.sqrt(6.0 / (1 + k))
        sd[name + ".weight"] = torch.from_numpy((rs.uniform(-1.0, 1.0, size=(o, k)) * a).astype(np.float32))
        sd[name + ".bias"] = torch.from_numpy(rs.uniform(-0.05, 0.05, size=(o,)).astype(np.float32))
    if like is not None:
        for k_, v in sd.items():
            assert tuple(like[k_].shape) == tuple(v.shape), (k_, like[k_].shape, v.shape)
    return sd


def load_seeded(graph, seed):
    """both networks of `graph` <- seeded_state(seed), seeded_state(seed + 1); progress and the pose network untouched"""
    for j, net in enumerate([graph.nerf] + ([graph.nerf_fine] if hasattr(graph, "nerf_fine") else [])):
        sd = seeded_state(seed + j)
        with torch.no_grad():
            for k, v in sd.items():
                mod, idx, what = k.split(".")
                getattr(getattr(net, mod)[int(idx)], what).copy_(v.to(getattr(getattr(net, mod)[int(idx)], what).device))
        if hasattr(net, "weights_changed"):
            net.weights_changed()


# ---------------------------------------------------------------------------------------------- recording
class TapedCalls:
    """wraps graph.render / graph.render_to_max: arguments, draws, outputs, upstream gradients, input gradients of every call"""

    def __init__(self, graph, tape):
        self.calls = []
        self.tape = tape
        for name in ("render", "render_to_max"):
            fn = getattr(graph, name)

            def wrapped(opt, pose, *a, __fn=fn, __name=name, **k):
                assert not a, "the reference passes everything after the pose by keyword (renderer.py:128-242, :498)"
                grad = torch.is_grad_enabled()
                rec = dict(method=__name, grad=grad, gout={}, gpose=None, gpix=None)
                cpu = lambda t: t.detach().float().cpu().clone()
                rec["pose"], rec["intr"] = cpu(pose), cpu(k["intr"])
                px, ix = k.get("pixels"), k.get("ray_idx")
                rec["pixels"] = cpu(px) if px is not None else None
                rec["ray_idx"] = ix.detach().cpu().clone() if ix is not None else None
                for key in ("depth_range", "depth_min", "depth_max"):
                    v = k.get(key)
                    if v is None:
                        rec[key] = None
                    elif torch.is_tensor(v):
                        rec[key] = ("tensor", cpu(v))
                    else:
                        rec[key] = ("list", [float(x) for x in v]) if isinstance(v, (list, tuple)) else ("float", float(v))
                rec["H"], rec["W"], rec["iter"], rec["mode"] = int(k["H"]), int(k["W"]), k.get("iter"), k.get("mode")
                if grad and pose.requires_grad:
                    pose = pose * 1.0                    # (exact) a node of this call's own: its hook sees this call's share of d loss / d pose
                    pose.register_hook(lambda g, r=rec: r.__setitem__("gpose", cpu(g)))
                if grad and px is not None and px.requires_grad:
                    px = px * 1.0
                    px.register_hook(lambda g, r=rec: r.__setitem__("gpix", cpu(g)))
                    k = dict(k, pixels=px)
                i0 = len(self.tape.log)
                ret = __fn(opt, pose, **k)
                rec["draws"] = list(zip(self.tape.log[i0:], self.tape.values[i0:]))
                rec["draw_range"] = (i0, len(self.tape.log))
                rec["out"] = {key: cpu(ret[key]) for key in OUT_KEYS if key in ret}
                # the merged (coarse + resampled, sorted) depths the fine pass was rendered at (renderer.py:334-336, returned as 't_fine'): lets a
                # replay FORCE them, so that the fine pass is compared on identical sample sets (`save` keeps them where the settings file asks
                # for density noise -- the inverse-CDF resampling is then ill-conditioned, see replay(force_fine_depths=))
                rec["t_fine"] = cpu(ret["t_fine"]) if (__name == "render" and "t_fine" in ret) else None
                for key in OUT_KEYS:
                    if key in ret and grad and ret[key].requires_grad:
                        ret[key].register_hook(lambda g, r=rec, kk=key: r["gout"].__setitem__(kk, cpu(g)))
                self.calls.append(rec)
                return ret

            setattr(graph, name, wrapped)


def record(name, seed, rays=4096, samples=(64, 128), device="cpu", scene_hw=None, iteration=110000, weight_seed=1000):
    """One iteration of the reference's loss code on the REFERENCE Graph -> tape (dict).  Needs the reference tree."""
    from tests import ref_harness as RH
    from easydict import EasyDict as edict
    opt = RH.load_settings(name, rays=rays, samples=samples, scene_hw=scene_hw)
    opt.device = str(device)
    scene = RH.make_scene(name, opt, device)
    torch.manual_seed(0)
    graph, opt = RH.build_graph("reference", opt, scene, device)
    load_seeded(graph, weight_seed)
    pose_state = {k: v.detach().cpu().clone() for k, v in graph.state_dict().items() if k.startswith("pose_net.")}
    tape = RH.DrawTape(seed=seed)
    from source.training.core.loss_factory import define_loss
    from source.training.core.sampling_strategies import RaySamplingStrategy
    with tape.run("record"):
        log = TapedCalls(graph, tape)
        loss_module = define_loss(opt.loss_type, opt, graph, scene.train_data, device, flow_net=scene.flow_net)
        sampler = RaySamplingStrategy(opt, data_dict=scene.train_data.all, device=device)
        data_dict = edict(scene.train_data.all)
        data_dict.iter = iteration
        progress = None
        if opt.barf_c2f is not None and opt.apply_cf_pe:                           # nerf_trainer.py:271-275
            progress = iteration / opt.max_iter
            graph.nerf.progress.data.fill_(progress)
            graph.nerf_fine.progress.data.fill_(progress)
        rays_idx = sampler(opt.nerf.rand_rays, sample_in_center=iteration < opt.precrop_iters)
        output_dict = graph.render_image_at_specific_rays(opt, data_dict, ray_idx=rays_idx, iter=iteration, mode="train")
        data_dict.poses_w2c = graph.get_w2c_pose(opt, data_dict, mode="train")
        loss_dict, _, _ = loss_module.compute_loss(opt, data_dict, output_dict, mode="train", plot=False, iteration=iteration)
        for p in graph.parameters():
            p.grad = None
        loss_dict["all"].backward()
    losses = {k: float(v.detach()) for k, v in loss_dict.items() if torch.is_tensor(v) and v.dim() == 0}
    grads = {n: p.grad.detach().float().cpu().clone() for n, p in graph.named_parameters() if p.grad is not None}
    return dict(name=name, seed=seed, weight_seed=weight_seed, iteration=iteration, progress=progress, opt=_jsonable(opt), calls=log.calls,
                draw_log=list(tape.log), draw_sums=[_draw_sum(v) for v in tape.values], losses=losses, grads=grads, pose_state=pose_state, rays=rays, samples=tuple(samples),
                scene_hw=scene_hw)


def _draw_sum(v):
    """checksum of one draw (float64 sum): `load` compares the regenerated stream against it"""
    return float(v.double().sum()) if torch.is_tensor(v) else float(np.asarray(v, dtype=np.float64).sum())


def _jsonable(o):
    if isinstance(o, dict):
        out = {}
        for k, v in o.items():
            if str(k).startswith("_"):
                continue
            j = _jsonable(v)
            if j is not _DROP:
                out[str(k)] = j
        return out
    if isinstance(o, (list, tuple)):
        items = [_jsonable(v) for v in o]
        return _DROP if any(i is _DROP for i in items) else items
    if isinstance(o, (bool, int, float, str)) or o is None:
        return o
    if isinstance(o, (np.integer, np.floating)):
        return o.item()
    return _DROP


_DROP = object()


def opt_from_json(d, precision=None):
    from sparf_amd.edict import EasyDict as edict

    def conv(x):
        return edict({k: conv(v) for k, v in x.items()}) if isinstance(x, dict) else x
    opt = conv(d)
    if precision is not None:
        opt.hip = edict(precision=precision)
    return opt


# ---------------------------------------------------------------------------------------------- replay
@contextlib.contextmanager
def inject_draws(draws, device):
    """torch.rand / randn / randn_like hand out this call's taped draws, FIFO per (kind, number of elements)"""
    fifo = {}
    for key, v in draws:
        fifo.setdefault((key[0], key[1]), []).append(v)
    real = (torch.rand, torch.randn, torch.randn_like)

    def take(kind, n, dev, shape):
        q = fifo.get((kind, n))
        if not q:
            raise AssertionError(f"replayed call asks for a draw ({kind}, {n}) the taped call never made (taped: {sorted(fifo)})")
        return q.pop(0).to(dev).reshape(shape)

    def shape_of(size):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        return tuple(int(s) for s in size)

    def rand(*size, **kw):
        shp = shape_of(size)
        return take("rand", int(np.prod(shp)), kw.get("device", "cpu"), shp)

    def randn(*size, **kw):
        shp = shape_of(size)
        return take("randn", int(np.prod(shp)), kw.get("device", "cpu"), shp)

    def randn_like(t, **kw):
        return take("randn", t.numel(), t.device, tuple(t.shape))

    torch.rand, torch.randn, torch.randn_like = rand, randn, randn_like
    try:
        yield fifo
    finally:
        torch.rand, torch.randn, torch.randn_like = real


def _arg(spec, device):
    if spec is None:
        return None
    kind, v = spec
    return v.to(device) if kind == "tensor" else v


def rel_max(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


@contextlib.contextmanager
def forced_fine_depths(opt, merged, device):
    """While active, a HIP `Graph.render` renders its fine pass at `merged` ([B, R, Nc + Nf, 1], the reference's own merged depth samples)
    instead of the depths it resamples from its own coarse weights: the render takes the pass-by-pass route (`opt.hip.fused_render =
    False`: same kernels, bit-identical to the fused route, tests/test_abi6_gpu.py) and the one call that produces the merged depths,
    `sparf_amd.ops.sample_fine`, is answered from the tape.  Test-side teacher forcing only: nothing in the product reads a tape."""
    from sparf_amd import ops
    hip = opt.get("hip", None)
    saved_flag = hip.get("fused_render", True) if hip is not None else True
    real = ops.sample_fine
    used = []

    def taped(weights, t_coarse, *a, **k):
        n, nc = t_coarse.shape
        m = merged.to(device).reshape(n, -1).contiguous()
        assert m.shape[1] > nc, "merged depths hold the coarse samples and the resampled ones"
        used.append(n)
        return m, None
    opt.hip.fused_render = False
    ops.sample_fine = taped
    try:
        yield used
    finally:
        ops.sample_fine = real
        opt.hip.fused_render = saved_flag


def replay(tape, graph, opt, device, force_fine_depths=False):
    """Drive `graph` (any renderer with the reference's Graph API) with the taped calls.  -> dict of error numbers:
    per_call[i] = {output key: max|a-b| / max|b|, 'd_pose': ..., 'd_pixels': ...}, grad_* over the accumulated parameter gradients.
    force_fine_depths: calls whose tape holds the reference's merged fine depths ('t_fine') render their fine pass AT those depths
    (HIP `Graph` only, `forced_fine_depths`); per_call[i]['_forced'] says which did.  Without it every call resamples from its own
    coarse weights, and `_t_fine` reports how far the two sample sets are apart."""
    if tape.get("progress") is not None:
        graph.nerf.progress.data.fill_(tape["progress"])
        graph.nerf_fine.progress.data.fill_(tape["progress"])
    for p in graph.parameters():
        p.grad = None
    per_call = []
    for c in tape["calls"]:
        pose = c["pose"].to(device).requires_grad_(c["gpose"] is not None)
        px = c["pixels"].to(device).requires_grad_(c["gpix"] is not None) if c["pixels"] is not None else None
        ix = c["ray_idx"].to(device) if c["ray_idx"] is not None else None
        kw = dict(intr=c["intr"].to(device), pixels=px, ray_idx=ix, mode=c["mode"], H=c["H"], W=c["W"], iter=c["iter"])
        if c["method"] == "render":
            kw["depth_range"] = _arg(c["depth_range"], device)
        else:
            kw["depth_min"], kw["depth_max"] = _arg(c["depth_min"], device), _arg(c["depth_max"], device)
        forced = bool(force_fine_depths and c.get("t_fine") is not None)
        with torch.set_grad_enabled(c["grad"]), inject_draws(c["draws"], device) as left, \
                (forced_fine_depths(opt, c["t_fine"], device) if forced else contextlib.nullcontext([])) as used:
            ret = getattr(graph, c["method"])(opt, pose, **kw)
        e = {k: rel_max(ret[k].detach().float().cpu().reshape(v.shape), v) for k, v in c["out"].items()}
        e["_forced"] = forced
        if forced:
            assert len(used) == 1, "the forced render asked for its merged depths exactly once"
        if c.get("t_fine") is not None and "t_fine" in ret:
            # the two renderers' merged sample sets, sample by sample: relative to the depth range
            mine, ref_t = ret["t_fine"].detach().float().cpu().reshape(c["t_fine"].shape).double(), c["t_fine"].double()
            span = float(ref_t.max() - ref_t.min()) + 1e-30
            d = (mine - ref_t).abs() / span
            e["_t_fine"] = dict(max=float(d.max()), mean=float(d.mean()), moved_gt_1e4=float((d > 1e-4).double().mean()), moved_gt_1e6=float((d > 1e-6).double().mean()))
        e["_missing_outputs"] = sorted(set(c["out"]) - set(ret.keys()))
        e["_unused_draws"] = {str(k): len(v) for k, v in left.items() if v}
        if c["gout"]:
            keys = sorted(c["gout"])
            torch.autograd.backward([ret[k] for k in keys], [c["gout"][k].to(device).reshape(ret[k].shape) for k in keys])
        if c["gpose"] is not None:
            e["d_pose"] = rel_max(pose.grad.detach().cpu(), c["gpose"]) if pose.grad is not None else float("inf")
        if c["gpix"] is not None:
            e["d_pixels"] = rel_l2(px.grad.detach().cpu(), c["gpix"]) if px.grad is not None else float("inf")
        per_call.append(e)
    out = dict(per_call=per_call, calls=[(c["method"], int(c["out"]["rgb"].shape[0] * c["out"]["rgb"].shape[1]) if "rgb" in c["out"] else None, c["grad"])
                                         for c in tape["calls"]])
    got = {n: p.grad.detach().float().cpu() for n, p in graph.named_parameters() if p.grad is not None and not n.startswith("pose_net.")}
    ref = {n: g for n, g in tape["grads"].items() if not n.startswith("pose_net.")}
    out["missing_grads"] = sorted(set(ref) - set(got))
    per = {}
    for n, g in ref.items():
        if n not in got:
            continue
        if isinstance(g, dict):                    # a loaded tape: a fixed subset of the entries + the norm of the whole tensor
            mine = got[n].reshape(-1)
            per[n] = dict(rel_l2=rel_l2(mine[g["idx"]], g["values"]), norm_ratio=float(mine.double().norm() / (g["norm"] + 1e-300)))
        else:
            per[n] = dict(rel_l2=rel_l2(got[n], g), norm_ratio=float(got[n].double().norm() / (g.double().norm() + 1e-300)))
    out["grad_per_tensor"] = per
    if per:
        out["grad_worst_name"] = max(per, key=lambda n: per[n]["rel_l2"])
        out["grad_worst_tensor"] = per[out["grad_worst_name"]]["rel_l2"]
        num = sum((per[n]["rel_l2"] * _ref_norm(ref[n])) ** 2 for n in per)
        out["grad_all"] = float(np.sqrt(num) / np.sqrt(sum(_ref_norm(ref[n]) ** 2 for n in per)))
        out["grad_norm_ratio_worst"] = max(abs(per[n]["norm_ratio"] - 1.0) for n in per)
    return out


def _ref_norm(g):
    """norm of the part of a reference gradient that rel_l2 was taken over (the kept subset of a loaded tape)"""
    return float(g["values"].double().norm()) if isinstance(g, dict) else float(g.double().norm())


# ---------------------------------------------------------------------------------------------- serialisation
def save(tape, path):
    """-> .npz: arrays + one JSON document.  Draws are NOT stored (regenerated from the seed by `load`), nor are the weights
    (`seeded_state`); gradient tensors above FULL_BELOW elements are kept as SUBSET entries at indices drawn from
    RandomState(seed of the tape) + the L2 norm of the whole tensor."""
    arrays, calls = {}, []
    for i, c in enumerate(tape["calls"]):
        m = {k: c[k] for k in ("method", "grad", "H", "W", "iter", "mode")}
        m["draw_range"] = list(c["draw_range"])
        for key in ("pose", "intr", "pixels", "ray_idx", "gpose", "gpix"):
            if c[key] is not None:
                arrays[f"c{i}.{key}"] = c[key].numpy()
        for key in ("depth_range", "depth_min", "depth_max"):
            if c[key] is None:
                m[key] = None
            elif c[key][0] == "tensor":
                m[key] = "tensor"
                arrays[f"c{i}.{key}"] = c[key][1].numpy()
            else:
                m[key] = list(c[key])
        for k, v in c["out"].items():
            arrays[f"c{i}.out.{k}"] = v.numpy()
        if c.get("t_fine") is not None and tape.get("keep_t_fine"):
            arrays[f"c{i}.t_fine"] = c["t_fine"].numpy()
        for k, v in c["gout"].items():
            arrays[f"c{i}.gout.{k}"] = v.numpy()
        m["out_keys"], m["gout_keys"] = sorted(c["out"]), sorted(c["gout"])
        calls.append(m)
    rs = np.random.RandomState(tape["seed"] + 77)
    gmeta = {}
    for n, g in tape["grads"].items():
        flat = g.reshape(-1)
        if n.startswith("pose_net.") or flat.numel() <= FULL_BELOW:
            arrays[f"g.{n}"] = g.numpy()
            gmeta[n] = dict(full=True)
        else:
            idx = np.sort(rs.choice(flat.numel(), SUBSET, replace=False)).astype(np.int64)
            arrays[f"g.{n}.idx"] = idx
            arrays[f"g.{n}.values"] = flat[torch.from_numpy(idx)].numpy()
            gmeta[n] = dict(full=False, norm=float(flat.double().norm()), numel=int(flat.numel()))
    for k, v in tape["pose_state"].items():
        arrays[f"p.{k}"] = v.numpy()
    meta = dict(name=tape["name"], seed=tape["seed"], weight_seed=tape["weight_seed"], iteration=tape["iteration"], progress=tape["progress"],
                opt=tape["opt"], calls=calls, draw_log=[[k[0], list(k[1]) if isinstance(k[1], tuple) else k[1]] for k in tape["draw_log"]],
                draw_sums=tape["draw_sums"], losses=tape["losses"], grads=gmeta, rays=tape["rays"], samples=list(tape["samples"]), scene_hw=tape["scene_hw"],
                torch=torch.__version__, numpy=np.__version__)
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(path, **arrays)
    return path


def regenerate_draws(seed, draw_log):
    """the values of every taped draw, in order, from the seed (the generators of ref_harness.DrawTape, same call sequence)"""
    from tests.ref_harness import DrawTape
    t = DrawTape(seed=seed)
    vals = []
    for kind, arg in draw_log:
        if kind == "rand":
            vals.append(t.np_uniform((arg,)))
        elif kind == "randn":
            vals.append(t.np_normal((arg,)))
        elif kind == "randperm":
            vals.append(t.np_perm(arg))
        elif kind == "np.rand":
            vals.append(float(t.rs.random_sample()) if not arg else t.rs.random_sample(tuple(arg)))
        elif kind == "np.randint":
            vals.append(t.rs.randint(*arg))
        else:
            raise ValueError(kind)
    return vals


def load(path):
    z = np.load(path)
    meta = json.loads(bytes(z["meta"]).decode())
    vals = regenerate_draws(meta["seed"], meta["draw_log"])
    keys = [(k, tuple(a) if isinstance(a, list) else a) for k, a in meta["draw_log"]]
    bad = [i for i, (v, s_) in enumerate(zip(vals, meta["draw_sums"])) if abs(_draw_sum(v) - s_) > 1e-6 * max(1.0, abs(s_))]
    if bad:
        raise RuntimeError(f"{path}: the draws regenerated from seed {meta['seed']} differ from the taped ones at {bad[:5]} (numpy {np.__version__}, "
                           f"tape made with {meta['numpy']}): RandomState's legacy streams are expected to be stable")
    T = lambda name: torch.from_numpy(z[name])
    calls = []
    for i, m in enumerate(meta["calls"]):
        c = {k: m[k] for k in ("method", "grad", "H", "W", "iter", "mode")}
        for key in ("pose", "intr", "pixels", "ray_idx", "gpose", "gpix"):
            c[key] = T(f"c{i}.{key}") if f"c{i}.{key}" in z else None
        for key in ("depth_range", "depth_min", "depth_max"):
            c[key] = None if m[key] is None else ("tensor", T(f"c{i}.{key}")) if m[key] == "tensor" else (m[key][0], m[key][1])
        c["out"] = {k: T(f"c{i}.out.{k}") for k in m["out_keys"]}
        c["t_fine"] = T(f"c{i}.t_fine") if f"c{i}.t_fine" in z else None
        c["gout"] = {k: T(f"c{i}.gout.{k}") for k in m["gout_keys"]}
        i0, i1 = m["draw_range"]
        c["draws"] = [(keys[j], vals[j]) for j in range(i0, i1)]
        c["draw_range"] = (i0, i1)
        calls.append(c)
    grads = {}
    for n, gm in meta["grads"].items():
        grads[n] = T(f"g.{n}") if gm["full"] else dict(idx=torch.from_numpy(z[f"g.{n}.idx"]), values=T(f"g.{n}.values"), norm=gm["norm"])
    pose_state = {k[2:]: T(k) for k in z.files if k.startswith("p.")}
    return dict(name=meta["name"], seed=meta["seed"], weight_seed=meta["weight_seed"], iteration=meta["iteration"], progress=meta["progress"],
                opt=meta["opt"], calls=calls, losses=meta["losses"], grads=grads, pose_state=pose_state, rays=meta["rays"],
                samples=tuple(meta["samples"]), made_with=dict(torch=meta["torch"], numpy=meta["numpy"]))
