"""Far rows of a pass (C ABI 4, include/sparf_hip.h): the last K samples of every ray run through a second precision's kernels.

The routing machinery is checked EXACTLY, with fp32 as both the main and the far precision: whatever row takes whichever launch,
the arithmetic per row is the same, so
  * per-sample and rendered outputs are bit-identical to the plain fp32 pass,
  * ray (pose) gradients are bit-identical (the far dgrad writes the same per-sample point gradients into the same rows),
  * parameter gradients agree to summation order (far rows are accumulated in their own split-K partials and added).
Then the mode it exists for -- bf16x3 main, fp32 far, inverse-depth samples out to t ~ 1e6 (renderer.py:413-416): far rows carry the
fp32 kernels' values bit for bit, the other rows the bf16x3 kernels', and the rendered outputs are closer to the float64 referee
than the all-bf16x3 pass.  The end-to-end bounds at BASELINE config 3 are tests/test_scale_gpu.py's."""
import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from sparf_amd import lib as L
from sparf_amd import ops
from tests.golden.recipe import small_opt, make_state_dict
from tests.test_hip_gpu import dev, make_scene, params_list, rel_err, rel_l2

pytestmark = pytest.mark.gpu


def _inputs(R, N, seed, pose=True):
    d = dev()
    opt = small_opt(nerf=dict(depth=dict(param="inverse", range=[1, 0])))
    sd = make_state_dict(opt, 11)
    center, dirs, jitter, _ = make_scene(R, N, seed)
    t = O.sample_depth(opt, 1, R, N, [1, 0], "train", jitter)[0, :, :, 0]          # inverse depth: t_{N-1} = N / (1 - u), the far sample
    rs = np.random.RandomState(100 + seed)
    lw = {k: torch.from_numpy(rs.uniform(-1, 1, size=s).astype(np.float32)).to(d) for k, s in
          (("rgb", (R, 3)), ("depth", (R,)), ("opacity", (R,)), ("weights", (R, N)))}
    lw["depth"] = lw["depth"] / t.max()
    return opt, sd, center, dirs, t, lw


def _run(prec, far, opt, sd, center, dirs, t, lw, grad=True, pose=True):
    d = dev()
    plist = [p.clone().requires_grad_(grad) for p in params_list(sd, d)]
    packed = ops.pack_weights(plist, prec)
    c2f = ops.c2f_weights(sd["progress"].to(d), opt.barf_c2f, d)
    cg, dg = center.to(d).requires_grad_(grad and pose), dirs.to(d).requires_grad_(grad and pose)
    farg = (far[0], far[1], ops.pack_weights(plist, far[1])) if far is not None else None
    with torch.set_grad_enabled(grad):
        out = ops.nerf_pass(cg, dg, t.to(d), None, 0.0, False, prec, packed, c2f, plist, far=farg)
        if grad:
            sum((out[k] * lw[k]).sum() for k in lw).backward()
    grads = dict(params=torch.cat([p.grad.reshape(-1) for p in plist]) if grad else None, center=cg.grad, dirs=dg.grad)
    return out, grads


@pytest.mark.parametrize("K", [1, 5, 8])
@pytest.mark.parametrize("R,N", [(70, 24), (333, 64)])
def test_routing_is_exact_when_both_precisions_are_fp32(R, N, K):
    args = _inputs(R, N, 3)
    plain, gp = _run(L.PREC_FP32, None, *args)
    routed, gr = _run(L.PREC_FP32, (K, L.PREC_FP32), *args)
    for k in plain:
        assert torch.equal(plain[k], routed[k]), (k, rel_err(routed[k], plain[k]))
    assert torch.equal(gp["center"], gr["center"]) and torch.equal(gp["dirs"], gr["dirs"])
    assert rel_l2(gr["params"], gp["params"]) < 2e-6, rel_l2(gr["params"], gp["params"])
    # and without ray gradients (the non-pose dgrad variant), inference kernels
    plain, gp = _run(L.PREC_FP32, None, *args, pose=False)
    routed, gr = _run(L.PREC_FP32, (K, L.PREC_FP32), *args, pose=False)
    assert rel_l2(gr["params"], gp["params"]) < 2e-6
    plain, _ = _run(L.PREC_FP32, None, *args, grad=False)
    routed, _ = _run(L.PREC_FP32, (K, L.PREC_FP32), *args, grad=False)
    for k in plain:
        assert torch.equal(plain[k], routed[k]), k


@pytest.mark.parametrize("K", [1, 8])
def test_bf16x3_with_fp32_far_rows(K):
    R, N = 333, 64
    opt, sd, center, dirs, t, lw = args = _inputs(R, N, 5)
    assert float(t[:, -1].max()) > 1e3 and float(t[:, -2].max()) < N + 1          # only the last sample leaves [1, N]
    x3, g3 = _run(L.PREC_X3, None, *args)
    f32, g32 = _run(L.PREC_FP32, None, *args)
    mix, gm = _run(L.PREC_X3, (K, L.PREC_FP32), *args)
    for k in ("density_samples", "rgb_samples"):
        assert torch.equal(mix[k][:, N - K:], f32[k][:, N - K:]), k              # far rows: the fp32 kernels' values, bit for bit
        assert torch.equal(mix[k][:, :N - K], x3[k][:, :N - K]), k                # the others: the bf16x3 kernels'
    # float64 referee on the same inputs (tests/scale_cases.py convention)
    d = dev()
    sd64 = {k: v.to(d) for k, v in sd.items()}
    ref = O.pass_fixed(opt, sd64, center.to(d)[None], dirs.to(d)[None], t.to(d)[None, :, :, None], mode="train", compute_dtype=torch.float64)
    err = lambda o: max(rel_err(o[k].reshape(ref[k].shape), ref[k]) for k in ("rgb", "depth", "opacity", "weights", "depth_var"))
    e3, em, e32 = err(x3), err(mix), err(f32)
    print(f"K={K}: rendered error vs float64: bf16x3 {e3:.1e}, routed {em:.1e}, fp32 {e32:.1e}")
    assert em <= 1e-4 and em <= e3 * 1.05
    # gradients: the routed backward is the bf16x3 one on the near rows plus the fp32 one on the far rows -- no further from the
    # fp32 pass's gradients than the all-bf16x3 backward is
    for key in ("params", "center", "dirs"):
        e_mix, e_x3 = rel_l2(gm[key], g32[key]), rel_l2(g3[key], g32[key])
        print(f"   d {key}: routed vs fp32 {e_mix:.1e}, bf16x3 vs fp32 {e_x3:.1e}")
        assert e_mix <= max(1.5 * e_x3, 2e-2), (key, e_mix, e_x3)
