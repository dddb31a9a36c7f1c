"""Far rows of a pass (C ABI 4, include/sparf_hip.h): the last K samples of every ray run through the fp32 forward kernels as well;
their raw density / colour replace the main launch's before compositing, and what that launch SAVED (layer inputs, ReLU masks) is
transplanted into the pass's save area, so that the ordinary backward differentiates the forward that was composited.

Checked exactly: from three forwards of the same bf16x3-sized problem through the C ABI -- plain bf16x3, plain fp32, bf16x3 with fp32
far rows -- (i) per-sample outputs: far rows carry the fp32 kernels' values bit for bit, the others the bf16x3 kernels'; (ii) the save
area, decoded with the layout algebra of csrc/layout.h: near rows hold what the plain bf16x3 pass saved, far rows hold the fp32 pass's
saved activations rounded to bf16 and the fp32 pass's mask words.  Then the gradients of the routed pass against the fp32 pass's, next
to the all-bf16x3 pass's distance.  End-to-end bounds at BASELINE config 3: tests/test_00_scale_gpu.py."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from sparf_amd import lib as L
from sparf_amd import ops
from tests.golden.recipe import small_opt, make_state_dict
from tests.test_hip_gpu import dev, make_scene, params_list, rel_err, rel_l2

pytestmark = pytest.mark.gpu

# csrc/layout.h: saved buffers (columns) of a row, in order; 32-row tile blocks [buffer][16-byte chunk][row][CH elements] + 9 mask KiB
SAVE_BUFS = [320, 256, 256, 256, 256, 256, 256, 288, 128]
SAVE_COLS = sum(SAVE_BUFS)


def _inputs(R, N, seed):
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


def _forward_save(prec, far, sd, center, dirs, t):
    """one training forward through the C ABI; -> (outputs, save area as a uint8 tensor)"""
    d = dev()
    lib = L.load()
    plist = params_list(sd, d)
    packed = ops.pack_weights(plist, prec)
    c2f = ops.c2f_weights(sd["progress"].to(d), None, d)
    farg = (far[0], far[1], ops.pack_weights(plist, far[1])) if far is not None else None
    a, out, save, keep = ops.build_pass_fwd(prec, center.to(d).contiguous(), dirs.to(d).contiguous(), t.to(d).contiguous(), None, 0.0, False,
                                            packed, c2f, True, far=farg)
    L.check(lib.sparf_pass_forward(ctypes.byref(a), L.stream_ptr(d)), "fwd")
    torch.cuda.synchronize()
    return out, save


def _decode(save, rows, fp32):
    """save area -> (X [rows, 2272] as float32 in canonical (buffer, half h, slot q) order, masks [tiles, 9, 64, 4] int32)"""
    eb, ch = (4, 4) if fp32 else (2, 8)
    tile_bytes = SAVE_COLS * 32 * eb + 9 * 1024
    ntiles = save.numel() // tile_bytes
    blocks = save[:ntiles * tile_bytes].view(ntiles, tile_bytes)
    cols, off = [], 0
    for C in SAVE_BUFS:
        raw = blocks[:, off * 32 * eb:(off + C) * 32 * eb].contiguous()
        vals = raw.view(torch.float32 if fp32 else torch.bfloat16).view(ntiles, C // ch, 32, ch).float()      # [tile][chunk][row][el]
        x = vals.permute(0, 2, 1, 3).reshape(ntiles * 32, C)                                                 # [row][pos]
        # pos -> (h, q): pos = (q // ch) * 2ch + h * ch + q % ch
        pos = torch.arange(C, device=save.device)
        q = (pos // (2 * ch)) * ch + pos % ch
        h = (pos // ch) % 2
        order = torch.argsort(h * (C // 2) + q)                 # canonical column = h * (C/2) + q
        cols.append(x[:, order])
        off += C
    X = torch.cat(cols, dim=1)[:rows]
    masks = blocks[:, SAVE_COLS * 32 * eb:].contiguous().view(torch.int32).view(ntiles, 9, 64, 4)
    return X, masks


@pytest.mark.parametrize("K", [1, 8])
@pytest.mark.parametrize("R,N", [(70, 24), (333, 64)])
def test_far_rows_outputs_and_transplanted_saves_are_exact(R, N, K):
    opt, sd, center, dirs, t, _ = _inputs(R, N, 3)
    o3, s3 = _forward_save(L.PREC_X3, None, sd, center, dirs, t)
    o32, s32 = _forward_save(L.PREC_FP32, None, sd, center, dirs, t)
    om, sm = _forward_save(L.PREC_X3, (K, L.PREC_FP32), sd, center, dirs, t)
    far = torch.zeros(R, N, dtype=torch.bool, device=dev())
    far[:, N - K:] = True
    for k in ("sigma_raw", "rgb_samples"):
        assert torch.equal(om[k][far], o32[k][far]), k                 # far rows: the fp32 kernels' values, bit for bit
        assert torch.equal(om[k][~far], o3[k][~far]), k                # the others: the bf16x3 kernels'
    rows = R * N
    X3, M3 = _decode(s3, rows, fp32=False)
    X32, M32 = _decode(s32, rows, fp32=True)
    Xm, Mm = _decode(sm, rows, fp32=False)
    fr = far.reshape(-1)
    assert torch.equal(Xm[~fr], X3[~fr])                                # near rows: what the plain bf16x3 pass saved
    assert torch.equal(Xm[fr], X32[fr].bfloat16().float())              # far rows: the fp32 forward's activations, rounded to bf16
    # masks: lane slots (row % 32 + 32 h) of tile row // 32, all nine buffers
    g = torch.arange(rows, device=dev())
    tile, slot = g // 32, g % 32
    for h in (0, 1):
        got, f32m, x3m = Mm[tile, :, slot + 32 * h], M32[tile, :, slot + 32 * h], M3[tile, :, slot + 32 * h]
        assert torch.equal(got[fr], f32m[fr]) and torch.equal(got[~fr], x3m[~fr])


@pytest.mark.parametrize("K", [1, 8])
def test_bf16x3_with_fp32_far_rows_gradients(K):
    R, N = 333, 64
    opt, sd, center, dirs, t, lw = args = _inputs(R, N, 5)
    assert float(t[:, -1].max()) > 1e3 and float(t[:, -2].max()) < N + 1          # only the last sample leaves [1, N]
    x3, g3 = _run(L.PREC_X3, None, *args)
    f32, g32 = _run(L.PREC_FP32, None, *args)
    mix, gm = _run(L.PREC_X3, (K, L.PREC_FP32), *args)
    # float64 referee on the same inputs (tests/scale_cases.py convention)
    d = dev()
    sd64 = {k: v.to(d) for k, v in sd.items()}
    ref = O.pass_fixed(opt, sd64, center.to(d)[None], dirs.to(d)[None], t.to(d)[None, :, :, None], mode="train", compute_dtype=torch.float64)
    err = lambda o: max(rel_err(o[k].reshape(ref[k].shape), ref[k]) for k in ("rgb", "depth", "opacity", "weights", "depth_var"))
    e3, em, e32 = err(x3), err(mix), err(f32)
    print(f"K={K}: rendered error vs float64: bf16x3 {e3:.1e}, routed {em:.1e}, fp32 {e32:.1e}")
    assert em <= 1e-4 and em <= e3 * 1.05
    # gradients: every row's backward is bf16x3 arithmetic, on the far rows with the fp32 forward's masks and activations -- no further
    # from the fp32 pass's gradients than the all-bf16x3 backward is
    for key in ("params", "center", "dirs"):
        e_mix, e_x3 = rel_l2(gm[key], g32[key]), rel_l2(g3[key], g32[key])
        print(f"   d {key}: routed vs fp32 {e_mix:.1e}, bf16x3 vs fp32 {e_x3:.1e}")
        assert e_mix <= max(1.5 * e_x3, 2e-2), (key, e_mix, e_x3)
    # inference (no save area, no transplant): the same outputs
    mix_i, _ = _run(L.PREC_X3, (K, L.PREC_FP32), *args, grad=False)
    for k in ("rgb", "depth", "weights", "density_samples", "rgb_samples"):
        assert torch.equal(mix_i[k], mix[k].detach()), k


def test_far_tiles_by_value_take_each_tile_exactly_once():
    """C ABI far_count = -1 (render_to_max passes under inverse depth, inference only): every 128-row tile whose largest depth sample
    exceeds the threshold carries the fp32 inference kernel's values, every other tile the bf16x3 kernel's -- bit for bit, none twice,
    none never."""
    R, N, thr = 301, 64, 8.0                           # 19 264 rows: a ragged last tile
    opt, sd, center, dirs, _, _ = _inputs(R, N, 9)
    rs = np.random.RandomState(4)
    dmax = torch.from_numpy(np.where(rs.uniform(size=R) < 0.3, rs.uniform(20.0, 2e4, size=R), rs.uniform(1.5, 7.5, size=R)).astype(np.float32))
    t = O.sample_depth_to_max(N, 1.0, dmax[None])[0, :, :, 0]                      # renderer.py:616-621, depth_min = depth.range[0] = 1
    args = (opt, sd, center, dirs, t, None)
    x3, _ = _run(L.PREC_X3, None, *args, grad=False)
    f32, _ = _run(L.PREC_FP32, None, *args, grad=False)
    mix, _ = _run(L.PREC_X3, (thr, L.PREC_FP32), *args, grad=False)
    rows = R * N
    tmax_tile = torch.nn.functional.pad(t.reshape(-1), (0, (-rows) % 128), value=float(t.reshape(-1)[-1])).reshape(-1, 128).max(dim=1).values
    far = (tmax_tile > thr).repeat_interleave(128)[:rows].reshape(R, N).to(dev())
    assert 0.1 < float(far.float().mean()) < 0.9
    for k in ("density_samples", "rgb_samples"):
        assert torch.equal(mix[k][far], f32[k][far]) and torch.equal(mix[k][~far], x3[k][~far]), k
    with pytest.raises(L.SparfError):                  # training passes take the last-K form, not this one
        _run(L.PREC_X3, (thr, L.PREC_FP32), opt, sd, center, dirs, t, {k: torch.zeros(1, device=dev()) for k in ()}, grad=True)
