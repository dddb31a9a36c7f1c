"""Host-side pieces of sparf_amd/ops.py that need no GPU: the arena plan of the fused render (ops.RenderFn) and the flat parameter
proxy (ops.FlatParams / NeRF.flat_params) -- autograd plumbing only; every FLOP of the path runs in the HIP kernels (tests -m gpu)."""
import pytest
import torch

from sparf_amd import ops
from sparf_amd import lib as L


@pytest.mark.parametrize("R,Nc,Nf", [(4096, 64, 128), (4095, 64, 128), (1, 8, 8), (37, 16, 0), (455 * 9, 64, 128)])
def test_render_plan_tiles_the_arena_without_overlap(R, Nc, Nf):
    plan = ops._plan(R, Nc, Nf)
    for k in (0, 1):                 # arena 0: per-ray results + band weights, arena 1: per-sample results
        spans = sorted((off, off + n, name) for name, (off, n, kk) in plan.off.items() if kk == k)
        for (a0, a1, na), (b0, b1, nb) in zip(spans, spans[1:]):
            assert a1 <= b0, (na, nb)
        assert spans[-1][1] <= plan.total[k] and all(off % 64 == 0 for off, _, _ in spans)
        assert sum(s for _, s in plan.order[k]) == plan.total[k]
    # the per-ray outputs a caller may keep alive (Graph.render_by_slices, logging) share no allocation with the per-sample block
    per_ray = {n for n, (_, _, k) in plan.off.items() if k == 0}
    assert {"crgb", "cdepth", "copacity", "cdepth_var", "crgb_var", "call_cumulated", "craylen", "cc2f"} <= per_ray
    assert not per_ray & {"cweights", "cdensity", "crgb_samples", "csigma_raw", "ct"}
    assert plan.total[0] <= 64 * (R // 64 + 1) * 10 * (2 if Nf else 1) + 128
    # what a pass writes per ray and per sample (include/sparf_hip.h sparf_pass_fwd_t)
    for tag, N in (("c", Nc),) + ((("f", Nc + Nf),) if Nf else ()):
        assert plan.off[tag + "rgb"][1] == 3 * R and plan.off[tag + "rgb_samples"][1] == 3 * R * N and plan.off[tag + "t"][1] == R * N
        assert plan.off[tag + "c2f"][1] == 16
    assert ("ft" in plan.off) == bool(Nf)
    assert ops._plan(R, Nc, Nf) is plan              # cached


def test_flat_params_routes_gradients_like_per_parameter_inputs():
    torch.manual_seed(0)
    params = [torch.randn(o, i, requires_grad=True) if k == 0 else torch.randn(o, requires_grad=True) for (o, i) in L.LAYER_SHAPES for k in (0, 1)]
    flat = ops.FlatParams.apply(*params)
    assert flat.numel() == L.N_PARAMS and sum(ops._PARAM_SIZES) == L.N_PARAMS
    w = torch.randn(L.N_PARAMS)
    (flat * w).sum().backward()
    off = 0
    for p in params:
        assert torch.equal(p.grad.reshape(-1), w[off:off + p.numel()]) and p.grad.shape == p.shape
        off += p.numel()
    # one flat buffer behind all twenty gradients: what optim.FusedAdam and parallel.GradBucket reduce in place
    from sparf_amd.parallel import _group_grads
    flats, loose = _group_grads([p.grad for p in params])
    assert len(flats) == 1 and not loose and flats[0].numel() == L.N_PARAMS
    # a second backward through the same proxy (two losses of one iteration) accumulates; autograd.grad works through it
    (flat * 2.0).sum().backward()
    assert torch.allclose(params[0].grad.reshape(-1), w[:params[0].numel()] + 2.0)
    g = torch.autograd.grad((flat * 3.0).sum(), params[:2])
    assert float(g[0][0, 0]) == 3.0 and float(g[1][0]) == 3.0


def test_flat_params_without_gradient_leaves_grad_none():
    class Two(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a, b):
            ctx.set_materialize_grads(False)
            return a.sum() * 1.0, b.sum() * 1.0

        @staticmethod
        def backward(ctx, ga, gb):
            return (torch.ones(6) * ga if ga is not None else None), None        # the second network's pass received no gradient

    a, b = torch.randn(2, 3, requires_grad=True), torch.randn(4, requires_grad=True)
    fa, fb = ops.FlatParams.apply(a), ops.FlatParams.apply(b)
    oa, ob = Two.apply(fa, fb)
    oa.backward()
    assert a.grad is not None and b.grad is None


def test_model_with_a_cached_flat_proxy_deep_copies_and_forgets_it_on_mode_change():
    """NeRF.flat_params caches a non-leaf autograd tensor on the module; copy.deepcopy of such a module raised, and the cache
    kept the last iteration's parameter route alive"""
    import copy
    from sparf_amd.config import default_opt
    from sparf_amd.frequency_nerf import NeRF
    net = NeRF(default_opt())
    flat = net.flat_params()
    assert flat.grad_fn is not None and net._flat is not None
    twin = copy.deepcopy(net)
    assert twin._flat is None and twin._packed == {}
    assert all(torch.equal(a, b) and a.data_ptr() != b.data_ptr() for a, b in zip(net.hip_params(), twin.hip_params()))
    assert net._flat is not None and net.flat_params() is flat          # the original keeps its cache
    net.eval()
    assert net._flat is None
    net.flat_params()
    net.weights_changed()
    assert net._flat is None


def test_pending_render_reads_launch_and_writes_do_not():
    """sparf_amd.renderer.PendingRender, the result object of a deferred render call: the EasyDict surface the reference's loss code uses
    (corres_loss.py:158-221: attribute reads, `'rgb_fine' in ret.keys()`; renderer.py:188: `ret.ray_idx = ...`) against a stub batch"""
    import copy
    from sparf_amd.renderer import PendingRender
    from sparf_amd.edict import EasyDict as edict

    class Stub:
        def __init__(self):
            self.results, self.flushed = [], 0

        def add(self):
            r = PendingRender(self)
            self.results.append(r)
            return r

        def flush(self):
            self.flushed += 1
            for i, r in enumerate(self.results):
                r._fill(dict(rgb=torch.full((2,), float(i)), depth=torch.zeros(1), ray_idx="from the render"))

    for read in (lambda r: r.rgb, lambda r: r["rgb"], lambda r: "rgb_fine" in r.keys(), lambda r: "rgb" in r, lambda r: list(r), lambda r: len(r),
                 lambda r: dict(r), lambda r: edict(r), lambda r: r.get("rgb"), lambda r: r.items(), lambda r: repr(r), lambda r: r.copy(), lambda r: {**r}):
        b = Stub()
        a, c = b.add(), b.add()
        a.ray_idx = "mine"                      # renderer.py:188, on a result that has not been rendered yet
        a["note"] = 1
        assert b.flushed == 0 and dict.__contains__(a, "ray_idx")
        assert not hasattr(a, "__getstate_manages_dict__") and not hasattr(a, "__torch_function__") and b.flushed == 0      # protocol probes are not reads
        read(c)                                 # reading EITHER result launches the whole batch, once
        assert b.flushed == 1
        assert float(a.rgb[0]) == 0.0 and float(c.rgb[0]) == 1.0 and b.flushed == 1
        assert a.ray_idx == "mine" and a.note == 1 and c.ray_idx == "from the render"        # what the caller wrote wins over the render's own key
        with pytest.raises(AttributeError):
            a.rgb_fine
        with pytest.raises(KeyError):
            a["rgb_fine"]
    # copies / pickles of a result that has not been read yet: plain EasyDicts of the rendered result
    import pickle
    for dup in (copy.copy, copy.deepcopy, lambda r: pickle.loads(pickle.dumps(r))):
        b = Stub()
        a = b.add()
        d = dup(a)
        assert b.flushed == 1 and type(d) is edict and float(d.rgb[0]) == 0.0 and not isinstance(d, PendingRender)
