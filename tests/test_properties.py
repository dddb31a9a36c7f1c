"""Property tests.  hypothesis draws shapes and values; the
properties are size-independent facts of the domain:

CPU (oracle + host logic, `-m "not gpu"`):
  * compositing: weights >= 0, sum to 1 (the 1e10 closing interval absorbs the rest),
    depth inside [t_0, t_last], rgb linear in the per-sample colours;
  * inverse-CDF resampling: samples inside the bin range, monotone in u, concentrated in the bin that
    carries the weight;
  * shard_slice partitions any ray count over any world size;
  * positional encoding: sin^2 + cos^2 = w_k^2 per band.
GPU (`-m gpu`): the HIP pass against the oracle for drawn ray / sample counts (ragged sizes around every
tile boundary), permutation equivariance over rays, invariance under ray-chunking, sortedness of the
merged fine depths.
"""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import nerf_oracle as O
from tests.golden.recipe import make_state_dict, small_opt

T = torch.from_numpy
CPU = dict(max_examples=25, deadline=None, derandomize=True, database=None, suppress_health_check=list(HealthCheck))      # same examples every run
GPU = dict(max_examples=12, deadline=None, derandomize=True, database=None, suppress_health_check=list(HealthCheck))


def _rays(rs, R):
    c = T(rs.uniform(-0.5, 0.5, size=(1, R, 3)).astype(np.float32)) + torch.tensor([0.0, 0.0, -3.0])
    r = T(rs.uniform(-0.3, 0.3, size=(1, R, 3)).astype(np.float32)) + torch.tensor([0.0, 0.0, 1.0])
    return c, r


@settings(**CPU)
@given(R=st.integers(1, 9), N=st.integers(2, 40), seed=st.integers(0, 10 ** 6), scale=st.floats(1e-3, 50.0))
def test_composite_invariants(R, N, seed, scale):
    rs = np.random.RandomState(seed)
    opt = small_opt()
    _, ray = _rays(rs, R)
    t = T(np.sort(rs.uniform(1.2, 5.2, size=(1, R, N, 1)), axis=2).astype(np.float32))
    dens = T((rs.gamma(0.7, 1.0, size=(1, R, N)) * scale).astype(np.float32))
    rgb_s = T(rs.uniform(size=(1, R, N, 3)).astype(np.float32))
    out = O.composite(opt, ray, rgb_s, dens, t)
    w = out["weights"][..., 0]
    assert float(w.min()) >= 0.0
    assert torch.allclose(w.sum(-1), torch.ones(1, R), atol=2e-5)                      # opacity == 1
    assert torch.all(out["depth"][..., 0] >= t[:, :, 0, 0] - 1e-4) and torch.all(out["depth"][..., 0] <= t[:, :, -1, 0] + 1e-4)
    assert float(out["all_cumulated"].max()) <= 1.0 + 1e-6
    # linear in the colours: composite(a c1 + b c2) = a composite(c1) + b composite(c2)
    c2 = T(rs.uniform(size=(1, R, N, 3)).astype(np.float32))
    mix = O.composite(opt, ray, 0.3 * rgb_s + 0.7 * c2, dens, t)["rgb"]
    assert torch.allclose(mix, 0.3 * out["rgb"] + 0.7 * O.composite(opt, ray, c2, dens, t)["rgb"], atol=1e-5)


@settings(**CPU)
@given(R=st.integers(1, 6), Nc=st.integers(2, 24), Nf=st.integers(1, 24), seed=st.integers(0, 10 ** 6), hot=st.integers(0, 23))
def test_sample_pdf_invariants(R, Nc, Nf, seed, hot):
    rs = np.random.RandomState(seed)
    w = T(rs.gamma(0.5, 1.0, size=(1, R, Nc)).astype(np.float32))
    grid = torch.sort(T(rs.uniform(size=Nf + 1).astype(np.float32))).values
    tf = O.sample_pdf(w, Nc, Nf, [1.2, 5.2], grid)[..., 0]
    assert float(tf.min()) >= 1.2 - 1e-5 and float(tf.max()) <= 5.2 + 1e-5
    assert torch.all(tf[..., 1:] >= tf[..., :-1] - 1e-5)                                 # sorted grid -> monotone samples
    # all the weight in one bin -> every sample inside that bin
    k = hot % Nc
    one = torch.zeros(1, 1, Nc)
    one[0, 0, k] = 1.0
    ts = O.sample_pdf(one, Nc, Nf, [1.2, 5.2], O.det_grid(Nf))[..., 0]
    lo, hi = 1.2 + 4.0 * k / Nc, 1.2 + 4.0 * (k + 1) / Nc
    assert float(ts.min()) >= lo - 1e-4 and float(ts.max()) <= hi + 1e-4


@settings(**CPU)
@given(n=st.integers(0, 10 ** 5), world=st.integers(1, 16))
def test_shard_slice_partition(n, world):
    from sparf_amd.parallel import shard_slice
    spans = [shard_slice(n, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1


@settings(**CPU)
@given(progress=st.floats(0.0, 1.0), L=st.sampled_from([4, 10]), seed=st.integers(0, 10 ** 6))
def test_encoding_band_energy(progress, L, seed):
    rs = np.random.RandomState(seed)
    opt = small_opt(barf_c2f=[0.4, 0.7])
    x = T(rs.uniform(-3, 3, size=(5, 3)).astype(np.float32))
    enc = O.positional_encoding(opt, x, L, torch.tensor(progress)).view(5, 3, 2, L)
    w = O.c2f_mask(opt, L, torch.tensor(progress))
    assert torch.allclose(enc[:, :, 0] ** 2 + enc[:, :, 1] ** 2, (w ** 2).expand(5, 3, L), atol=2e-5)
    assert torch.all(w[1:] <= w[:-1] + 1e-7) and float(w.min()) >= 0 and float(w.max()) <= 1      # coarse bands open first


# ------------------------------------------------------------------------------------------ GPU
def _pass(prec_name, opt, sd, c, r, t):
    from sparf_amd import lib as L, ops
    d = torch.device("cuda:0")
    plist = [sd[f"{n}.{k}"].to(d) for n in L.PARAM_NAMES for k in ("weight", "bias")]
    P = L.PREC_IDS[prec_name]
    packed = ops.pack_weights(plist, P)
    c2f = ops.c2f_weights(sd["progress"].to(d), opt.barf_c2f, d)
    return ops.nerf_pass(c.to(d), r.to(d), t.to(d), None, 0.0, False, P, packed, c2f, plist)


@pytest.mark.gpu
@settings(**GPU)
@given(R=st.integers(1, 300), N=st.integers(2, 70), seed=st.integers(0, 10 ** 6), prec=st.sampled_from(["fp32", "bf16x3"]))
def test_gpu_pass_matches_oracle_for_drawn_sizes(R, N, seed, prec):
    rs = np.random.RandomState(seed)
    opt = small_opt()
    sd = make_state_dict(opt, seed % 97)
    c, r = _rays(rs, R)
    t = T(np.sort(rs.uniform(1.2, 5.2, size=(1, R, N, 1)), axis=2).astype(np.float32))
    with torch.no_grad():
        got = _pass(prec, opt, sd, c[0], r[0], t[0, :, :, 0])
        ref = O.pass_fixed(opt, sd, c, r, t, mode="val")
    for k in ("rgb", "depth", "opacity", "weights", "all_cumulated"):
        a, b = got[k].cpu().reshape(-1).double(), ref[k].reshape(-1).double()
        assert float((a - b).abs().max()) <= 1e-4 * float(b.abs().max() + 1e-30), (k, R, N)


@pytest.mark.gpu
@settings(**GPU)
@given(R=st.integers(2, 200), N=st.integers(2, 48), seed=st.integers(0, 10 ** 6), split=st.integers(1, 199))
def test_gpu_rays_are_independent(R, N, seed, split):
    """Permuting the rays permutes the outputs bit for bit, and rendering a batch in two pieces gives the
    same rows as rendering it at once (what ray-batch sharding and render_batch rely on)."""
    rs = np.random.RandomState(seed)
    opt = small_opt()
    sd = make_state_dict(opt, 5)
    c, r = _rays(rs, R)
    t = T(np.sort(rs.uniform(1.2, 5.2, size=(R, N)), axis=1).astype(np.float32))
    perm = torch.from_numpy(rs.permutation(R))
    k = 1 + split % (R - 1)
    with torch.no_grad():
        full = _pass("fp32", opt, sd, c[0], r[0], t)
        shuf = _pass("fp32", opt, sd, c[0][perm], r[0][perm], t[perm])
        a, b = _pass("fp32", opt, sd, c[0][:k], r[0][:k], t[:k]), _pass("fp32", opt, sd, c[0][k:], r[0][k:], t[k:])
    for key in ("rgb", "depth", "weights"):
        assert torch.equal(shuf[key], full[key][perm.to(full[key].device)]), key
        assert torch.equal(torch.cat([a[key], b[key]]), full[key]), key


@pytest.mark.gpu
@settings(**GPU)
@given(R=st.integers(1, 120), Nc=st.integers(2, 64), Nf=st.integers(1, 128), seed=st.integers(0, 10 ** 6))
def test_gpu_merged_depths_sorted_and_complete(R, Nc, Nf, seed):
    from sparf_amd import ops
    rs = np.random.RandomState(seed)
    d = torch.device("cuda:0")
    w = T(rs.gamma(0.5, 1.0, size=(R, Nc)).astype(np.float32))
    tc = T(np.sort(rs.uniform(1.2, 5.2, size=(R, Nc)), axis=1).astype(np.float32))
    u = T(rs.uniform(size=Nf).astype(np.float32))
    merged, tf = ops.sample_fine(w.to(d), tc.to(d), u.to(d), 1.2, 5.2, want_unsorted=True)
    assert merged.shape == (R, Nc + Nf) and torch.all(merged[:, 1:] >= merged[:, :-1])
    both = torch.cat([tc.to(d), tf], dim=1).sort(dim=1).values                         # same multiset: the coarse depths survive bit for bit
    assert torch.equal(both, merged)
    assert float(tf.min()) >= 1.2 - 1e-5 and float(tf.max()) <= 5.2 + 1e-5


def _segments_case(R, N, seed, cuts, active, pose, far, q8, x3=False):
    from sparf_amd import lib as L, ops
    rs = np.random.RandomState(seed)
    opt = small_opt()
    sd = make_state_dict(opt, 3)
    d = torch.device("cuda:0")
    prec = L.PREC_X3 if (far or x3) else (L.PREC_X3 | L.SAVE_Q8) if q8 else L.PREC_FP32     # far rows exist for the bf16-plane modes (fp32 far rows under an fp32 pass would be the pass itself)
    tol = 3e-4 if (far or q8 or x3) else 2e-5                 # bf16x3: other split boundaries of the weight-gradient sums round differently
    c, r = _rays(rs, R)
    t = T(np.sort(rs.uniform(1.2, 5.2, size=(R, N)), axis=1).astype(np.float32)).to(d)
    bounds = sorted({0, R} | {x % R for x in cuts})
    segs = [(a, b - a, 0.0) for a, b in zip(bounds, bounds[1:])]
    on = [bool((active >> i) & 1) for i in range(len(segs))]
    if not any(on):
        on[-1] = True
    g_rgb, g_depth = T(rs.uniform(-1, 1, size=(R, 3)).astype(np.float32)).to(d), T(rs.uniform(-1, 1, size=(R,)).astype(np.float32)).to(d)
    g_w = T(rs.uniform(-1, 1, size=(R, N)).astype(np.float32)).to(d)

    def run(segmented):
        plist = [sd[f"{n}.{k}"].to(d).clone().requires_grad_(True) for n in L.PARAM_NAMES for k in ("weight", "bias")]
        packed = ops.pack_weights(plist, prec)
        c2f = ops.c2f_weights(sd["progress"].to(d), None, d)
        cg, dg = c[0].to(d).requires_grad_(pose), r[0].to(d).requires_grad_(pose)
        fr = (far, L.PREC_FP32, ops.pack_weights(plist, L.PREC_FP32)) if far else None
        if segmented:
            outs = ops.nerf_pass_segments(cg, dg, t, None, False, prec, packed, c2f, plist, segs, far=fr)
            loss = sum((o["rgb"] * g_rgb[a:a + n]).sum() + (o["depth"] * g_depth[a:a + n]).sum() + (o["weights"] * g_w[a:a + n]).sum()
                       for o, (a, n, _), use in zip(outs, segs, on) if use and n > 0)
            rgb = torch.cat([o["rgb"] for o in outs])
        else:
            o = ops.nerf_pass(cg, dg, t, None, 0.0, False, prec, packed, c2f, plist, far=fr)
            m = torch.zeros(R, device=d)
            for (a, n, _), use in zip(segs, on):
                m[a:a + n] = float(use)
            loss = (o["rgb"] * g_rgb * m[:, None]).sum() + (o["depth"] * g_depth * m).sum() + (o["weights"] * g_w * m[:, None]).sum()
            rgb = o["rgb"]
        loss.backward()
        return rgb.detach(), torch.cat([p.grad.reshape(-1) for p in plist]), (cg.grad, dg.grad) if pose else None

    rgb_s, gp_s, ray_s = run(True)
    rgb_p, gp_p, ray_p = run(False)
    assert torch.equal(rgb_s, rgb_p)
    assert float((gp_s - gp_p).abs().max()) <= tol * float(gp_p.abs().max() + 1e-30), (segs, on)
    if pose:
        for a, b in zip(ray_s, ray_p):
            assert float((a - b).abs().max()) <= tol * float(b.abs().max() + 1e-30)




@pytest.mark.gpu
@settings(**GPU)
@given(R=st.integers(3, 160), N=st.sampled_from([8, 32, 40, 64]), seed=st.integers(0, 10 ** 6), cuts=st.lists(st.integers(0, 159), min_size=1, max_size=5),
       active=st.integers(1, 62), pose=st.booleans(), far=st.sampled_from([0, 0, 1, 5]), q8=st.booleans())
def test_gpu_ray_segments_equal_zeroed_gradients(R, N, seed, cuts, active, pose, far, q8):
    """Ray segments of a pass (include/sparf_hip.h sparf_segment_t): for ANY partition of the rays into segments and ANY
    subset of segments carrying upstream gradients, the segmented backward -- which runs its kernels over the active ray
    range only, aligned or not to the 32-row tiles -- equals the plain backward fed the same gradients with zeros on the
    inactive rays; the forward is untouched by the table.  N = 32 / 64 with an inactive first segment is the tile-aligned
    `row_begin > 0` path of the backward kernels; `far` > 0 adds far rows (C ABI 4: the last `far` samples of every
    ray through the fp32 forward kernels, their saved activations and masks transplanted into the pass's save area) under a bf16x3 pass;
    `q8` (without far rows): the bf16x3 pass with 8-bit save / gradient areas (C ABI 5)."""
    _segments_case(R, N, seed, cuts, active, pose, far, q8)


@pytest.mark.gpu
@pytest.mark.parametrize("R,first,pose", [(1600, 70, True), (1600, 64, False), (1100, 1, True), (600, 88, False)])
def test_gpu_ray_segments_at_the_row_counts_where_the_data_gradient_kernel_changes_geometry(R, first, pose):
    """The same property at the sizes where the bf16x3 data-gradient launch is planned per row count (api.hip x3_dgrad_rows8, round 6):
    64 samples per ray with the first `first` rays inactive leave an active range that starts inside the pass (`row_begin` > 0, on a
    256-row tile boundary or not) and is, in turn, 1.5 rounds of 256-row tiles (the full round in 8 waves + the remainder in 4: two
    launches that must meet exactly at row_begin + 65 536), one-and-a-bit rounds, and less than one round (all in 4 waves)."""
    _segments_case(R, 64, 11, [first], 0b10, pose, 0, False, x3=True)


@pytest.mark.gpu
@pytest.mark.parametrize("prec,tol", [("fp32", 1e-5), ("bf16x3", 2e-2)])
def test_gpu_full_size_properties(prec, tol):
    """BASELINE configs[1]'s fine pass at full size (4096 rays x 192 samples = 786 432 rows, 24 tiles per CU: beyond what the
    oracle finishes in seconds), through properties that do not need it: the compositing invariants, ray-permutation
    equivariance bit for bit, and LINEARITY of the backward in the upstream gradients -- backward(a g1 + b g2) =
    a backward(g1) + b backward(g2) for parameters and rays, from one forward (the saved activations and masks are
    read-only: a second backward of the same graph is legal).  fp32 mode: to rounding; bf16x3: to the bf16 rounding of
    the propagated gradient (its heads are the dgrad / wgrad operands)."""
    from sparf_amd import lib as L, ops
    R, N = 4096, 192
    rs = np.random.RandomState(11)
    opt = small_opt()
    sd = make_state_dict(opt, 7)
    d = torch.device("cuda:0")
    P = L.PREC_IDS[prec]
    c, r = _rays(rs, R)
    t = T(np.sort(rs.uniform(1.2, 5.2, size=(R, N)), axis=1).astype(np.float32)).to(d)
    plist = [sd[f"{n}.{k}"].to(d).clone().requires_grad_(True) for n in L.PARAM_NAMES for k in ("weight", "bias")]
    packed = ops.pack_weights(plist, P)
    c2f = ops.c2f_weights(sd["progress"].to(d), None, d)
    cg, dg = c[0].to(d).requires_grad_(True), r[0].to(d).requires_grad_(True)
    o = ops.nerf_pass(cg, dg, t, None, 0.0, False, P, packed, c2f, plist)
    # forward invariants (frequency_nerf.py:283-343): weights >= 0, opacity = sum of weights <= 1, colours in [0, 1],
    # depth = sum w t inside [opacity t_first, opacity t_last]
    w, op = o["weights"].detach(), o["opacity"].detach().reshape(R)
    assert float(w.min()) >= 0 and float((w.sum(1) - op).abs().max()) <= 2e-5 and float(op.max()) <= 1 + 1e-5
    assert float(o["rgb"].detach().min()) >= -1e-6 and float(o["rgb"].detach().max()) <= 1 + 1e-5
    dep = o["depth"].detach().reshape(R)
    assert torch.all(dep >= op * t[:, 0] - 1e-4) and torch.all(dep <= op * t[:, -1] + 1e-4)
    perm = torch.from_numpy(rs.permutation(R)).to(d)
    with torch.no_grad():
        sh = ops.nerf_pass(cg.detach()[perm].contiguous(), dg.detach()[perm].contiguous(), t[perm].contiguous(), None, 0.0, False, P, packed, c2f, plist)
    for k in ("rgb", "depth", "opacity", "weights"):
        assert torch.equal(sh[k], o[k].detach()[perm]), k
    del sh

    def draw():
        return [T(rs.uniform(-1, 1, size=tuple(o[k].shape)).astype(np.float32)).to(d) for k in ("rgb", "depth", "opacity", "weights")]

    def grads(g):
        loss = sum((o[k] * gi).sum() for k, gi in zip(("rgb", "depth", "opacity", "weights"), g))
        out = torch.autograd.grad(loss, plist + [cg, dg], retain_graph=True)
        return torch.cat([x.reshape(-1) for x in out[:-2]]).double(), out[-2].double(), out[-1].double()

    g1, g2 = draw(), draw()
    a, b = 0.7, -1.3
    G1, G2, G3 = grads(g1), grads(g2), grads([a * x + b * y for x, y in zip(g1, g2)])
    for name, x1, x2, x3 in zip(("parameters", "d_center", "d_dir"), G1, G2, G3):
        want = a * x1 + b * x2
        err = float((x3 - want).norm() / (want.norm() + 1e-300))
        print(f"linearity[{prec}] {name}: rel l2 {err:.2e}")
        assert err <= tol, (name, err)
