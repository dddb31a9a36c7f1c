"""8-bit save / gradient areas (C ABI 5, include/sparf_hip.h SPARF_SAVE_Q8; csrc/layout.h AREA_Q8): a training pass of a bf16-operand
mode keeps its layer inputs, and hands the weight-gradient kernel its pre-activation gradients, as 8-bit integers on a linear grid
with one fp32 step per sample row and vector.

Checked EXACTLY, through the C ABI, against the plane format of the same mode (whose forward and data-gradient arithmetic the
8-bit format shares): (i) every output of the forward, d_center and d_dir: bit-identical; (ii) the save area and the gradient area,
decoded with the layout algebra of csrc/layout.h: byte for byte the quantisation u = rne(x * (127 / max|x|)) + 128 of what the
plane format holds, steps = max|x| / 127, mask words unchanged.  Then the weight gradients, whose operands these are: within the
quantiser's error of the plane format's (the measured level is DESIGN.md 6; the bound here is 3x that)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from sparf_amd import lib as L
from sparf_amd import ops
from tests.golden.recipe import small_opt, make_state_dict
from tests.test_hip_gpu import dev, make_scene, params_list

pytestmark = pytest.mark.gpu

SAVE_BUFS = [320, 256, 256, 256, 256, 256, 256, 288, 128]                 # csrc/layout.h SaveBuf
GRAD_BUFS = [256] * 7 + [288, 128, 32]                                   # csrc/layout.h GradBuf


def _canon(x, C, ch):
    """[rows][pos] -> canonical column = h * (C / 2) + q, pos = (q // ch) * 2 ch + h * ch + q % ch (layout.h pos_of)"""
    pos = torch.arange(C, device=x.device)
    q = (pos // (2 * ch)) * ch + pos % ch
    h = (pos // ch) % 2
    return x[:, torch.argsort(h * (C // 2) + q)]


def decode_planes(area, bufs, n_mask_kib):
    """bf16 plane area -> ([rows_padded, sum(bufs)] float32 in canonical order per buffer, mask bytes per tile)"""
    cols = sum(bufs)
    tile_bytes = cols * 32 * 2 + n_mask_kib * 1024
    ntiles = area.numel() // tile_bytes
    blocks = area[:ntiles * tile_bytes].view(ntiles, tile_bytes)
    out, off = [], 0
    for C in bufs:
        raw = blocks[:, off * 64:(off + C) * 64].contiguous()
        vals = raw.view(torch.bfloat16).view(ntiles, C // 8, 32, 8).float()                  # [tile][chunk][row][el]
        out.append(_canon(vals.permute(0, 2, 1, 3).reshape(ntiles * 32, C), C, 8))
        off += C
    return torch.cat(out, dim=1), blocks[:, cols * 64:cols * 64 + n_mask_kib * 1024]


def decode_q8(area, bufs, n_mask_kib):
    """8-bit area -> (u [rows_padded, sum(bufs)] int32 canonical, steps [rows_padded, len(bufs), 2] float32, mask bytes per tile)"""
    cols = sum(bufs)
    tile_bytes = cols * 32 + n_mask_kib * 1024 + len(bufs) * 256
    ntiles = area.numel() // tile_bytes
    blocks = area[:ntiles * tile_bytes].view(ntiles, tile_bytes)
    out, off = [], 0
    for C in bufs:
        raw = blocks[:, off * 32:(off + C) * 32].contiguous().view(ntiles, C // 32, 2, 32, 16)     # [tile][block][h][row][slot in block]
        u = raw.permute(0, 3, 2, 1, 4).reshape(ntiles * 32, C).int()                               # [row][h][block][slot] = h * C/2 + q
        out.append(u)
        off += C
    so = cols * 32 + n_mask_kib * 1024
    steps = blocks[:, so:so + len(bufs) * 256].contiguous().view(torch.float32).view(ntiles, len(bufs), 2, 32)
    steps = steps.permute(0, 3, 1, 2).reshape(ntiles * 32, len(bufs), 2)
    return torch.cat(out, dim=1), steps, blocks[:, cols * 32:so]


def encode_planes(X, bufs, tail):
    """inverse of decode_planes: canonical-order values [rows_padded, sum(bufs)] -> bf16 plane area bytes; `tail`: the bytes that follow
    the planes in every tile block (mask words), [ntiles, n] uint8"""
    ntiles = X.shape[0] // 32
    parts, off = [], 0
    for C in bufs:
        pos = torch.arange(C, device=X.device)
        q = (pos // 16) * 8 + pos % 8
        h = (pos // 8) % 2
        order = torch.argsort(h * (C // 2) + q)
        x = torch.empty(X.shape[0], C, device=X.device)
        x[:, order] = X[:, off:off + C]
        vals = x.view(ntiles, 32, C // 8, 8).permute(0, 2, 1, 3).contiguous().to(torch.bfloat16)
        parts.append(vals.view(torch.uint8).reshape(ntiles, C * 64))
        off += C
    return torch.cat(parts + [tail], dim=1).reshape(-1)


def dequantise(U, S, bufs):
    """what the weight-gradient kernel multiplies out: bf16(fma(u, step, -128 step)), canonical order, as float32"""
    out, off = [], 0
    for b, C in enumerate(bufs):
        half = C // 2
        part = ((torch.arange(C, device=U.device) % half) >= 128).long()
        st = S[:, b, :][:, part].double()                                                # [rows, C]
        v = (U[:, off:off + C].double() * st - 128.0 * st).float()                       # exact in float64, one rounding to float32 = the FMA
        out.append(v.to(torch.bfloat16).float())
        off += C
    return torch.cat(out, dim=1)


def quantise(X, bufs):
    """the quantiser of mlp_dev.h "8-bit saves" on canonical-order plane values: -> (u int32, steps [rows, nbuf, 2] float32)"""
    us, steps, off = [], [], 0
    for C in bufs:
        x = X[:, off:off + C]
        half = C // 2
        q = torch.arange(C, device=X.device) % half
        u = torch.empty_like(x, dtype=torch.int32)
        st = torch.zeros(x.shape[0], 2, device=X.device)
        for part, sel in ((0, q < 128), (1, q >= 128)):                       # part 1: the columns from 256 on (slots q >= 128 of both halves)
            if not bool(sel.any()):
                continue
            xs = x[:, sel]
            amax = xs.abs().amax(dim=1)
            f = torch.where(amax > 0, torch.tensor(127.0, device=X.device) / amax, torch.zeros_like(amax))      # fp32 division
            u[:, sel] = (torch.round(xs.double() * f.double()[:, None]) + 128).int()                            # rne of the exact product
            st[:, part] = amax * np.float32(1.0 / 127.0)
        us.append(u)
        steps.append(st)
        off += C
    return torch.cat(us, dim=1), torch.stack(steps, dim=1)


def _same(got, want, what, X=None):
    """exact equality with a useful report: how many entries differ, and the first few (column, value, got, want)"""
    if torch.equal(got, want):
        return
    bad = (got != want).nonzero()
    msg = [f"{what}: {bad.shape[0]} of {got.numel()} entries differ"]
    for r, c in bad[:8].tolist():
        msg.append(f"  row {r} col {c}: got {int(got[r, c])} want {int(want[r, c])}" + (f" x {float(X[r, c])!r}" if X is not None else ""))
    cols = torch.unique(bad[:, 1])
    msg.append(f"  columns affected: {cols[:32].tolist()} ({cols.numel()} distinct); rows affected: {torch.unique(bad[:, 0]).numel()}")
    raise AssertionError("\n".join(msg))


def _inputs(R, N, seed):
    d = dev()
    opt = small_opt()
    sd = make_state_dict(opt, 11)
    center, dirs, jitter, _ = make_scene(R, N, seed)
    t = O.sample_depth(opt, 1, R, N, opt.nerf.depth.range, "train", jitter)[0, :, :, 0]
    rs = np.random.RandomState(100 + seed)
    g = [torch.from_numpy(rs.uniform(-1, 1, size=s).astype(np.float32)).to(d) for s in ((R, 3), (R,), (R,), (R, N))]
    return opt, sd, center.to(d).contiguous(), dirs.to(d).contiguous(), t.to(d).contiguous(), g


def _pass(prec, opt, sd, center, dirs, t, grads, pose):
    """forward + backward of one pass through the C ABI -> (outputs, save area, workspace (gradient area first), parameter gradient, d_center, d_dir)"""
    d = dev()
    lib = L.load()
    plist = params_list(sd, d)
    packed = ops.pack_weights(plist, prec)
    c2f = ops.c2f_weights(sd["progress"].to(d), opt.barf_c2f, d)
    a, out, save, keep = ops.build_pass_fwd(prec, center, dirs, t, None, 0.0, False, packed, c2f, True)
    L.check(lib.sparf_pass_forward(ctypes.byref(a), L.stream_ptr(d)), "fwd")
    b, gp, dc, dd, keep2 = ops.build_pass_bwd(prec, center, dirs, t, None, 0.0, False, packed, c2f, save, out, tuple(grads), pose)
    L.check(lib.sparf_pass_backward(ctypes.byref(b), L.stream_ptr(d)), "bwd")
    torch.cuda.synchronize()
    ws = next(k for k in keep2 if isinstance(k, torch.Tensor) and k.dtype == torch.uint8 and k.data_ptr() == b.ws)
    return out, save, ws, gp, dc, dd


@pytest.mark.parametrize("base", ["bf16", "bf16x3"])
@pytest.mark.parametrize("R,N,pose", [(70, 24, False), (333, 64, True), (1024, 32, True)])
def test_q8_areas_are_the_quantised_plane_areas(base, R, N, pose):
    opt, sd, center, dirs, t, g = _inputs(R, N, 5)
    rows = R * N
    o0, s0, w0, gp0, dc0, dd0 = _pass(L.PREC_IDS[base], opt, sd, center, dirs, t, g, pose)
    o1, s1, w1, gp1, dc1, dd1 = _pass(L.PREC_IDS[base + "+q8"], opt, sd, center, dirs, t, g, pose)
    for k in o0:
        assert torch.equal(o0[k], o1[k]), k                                    # the forward is the plain mode's, bit for bit
    if pose:
        assert torch.equal(dc0, dc1) and torch.equal(dd0, dd1)                 # and so is the data-gradient chain
    assert s1.numel() < 0.56 * s0.numel()
    # save area
    X, M = decode_planes(s0, SAVE_BUFS, 9)
    U, S, Mq = decode_q8(s1, SAVE_BUFS, 9)
    Ue, Se = quantise(X, SAVE_BUFS)
    nt32 = (rows + 31) // 32                                                   # (tiles past the last row are padding: never written by the 128-row workgroups of bf16x3)
    assert torch.equal(M[:nt32], Mq[:nt32])                                    # ReLU mask words
    _same(U[:rows], Ue[:rows], "save area", X[:rows])
    used = torch.tensor([[1, 1]] + [[1, 0]] * 6 + [[1, 1], [1, 0]], dtype=torch.bool, device=dev())      # XS and FV have a second vector
    assert torch.equal(S[:rows][:, used], Se[:rows][:, used])
    # gradient area (first in the workspace, sparf_bwd_workspace_bytes)
    lib = L.load()
    nt = (rows + 255) // 256 * 8
    G, _ = decode_planes(w0[:nt * sum(GRAD_BUFS) * 64], GRAD_BUFS, 0)
    Ug, Sg, _ = decode_q8(w1[:nt * (sum(GRAD_BUFS) * 32 + len(GRAD_BUFS) * 256)], GRAD_BUFS, 0)
    Uge, Sge = quantise(G, GRAD_BUFS)
    _same(Ug[:rows], Uge[:rows], "gradient area", G[:rows])
    usedg = torch.tensor([[1, 0]] * 7 + [[1, 1], [1, 0], [1, 0]], dtype=torch.bool, device=dev())         # DY7: the raw-density slot is a vector of its own
    assert torch.equal(Sg[:rows][:, usedg], Sge[:rows][:, usedg])
    # weight gradients, exactly: the PLANE kernel on plane areas that hold the dequantised operands computes the same sums over the
    # same bf16 values in the same order
    Xd, Gd = dequantise(U, S, SAVE_BUFS), dequantise(Ug, Sg, GRAD_BUFS)
    s2 = encode_planes(Xd, SAVE_BUFS, Mq)
    assert s2.numel() <= s0.numel()
    s0[:s2.numel()] = s2
    g2 = encode_planes(Gd, GRAD_BUFS, torch.empty(Gd.shape[0] // 32, 0, dtype=torch.uint8, device=dev()))
    w0[:g2.numel()] = g2
    d = dev()
    plist = params_list(sd, d)
    prec0 = L.PREC_IDS[base]
    packed = ops.pack_weights(plist, prec0)
    c2f = ops.c2f_weights(sd["progress"].to(d), opt.barf_c2f, d)
    fa, _, _, keep = ops.build_pass_fwd(prec0, center, dirs, t, None, 0.0, False, packed, c2f, False)
    ba, gp2, _, _, keep2 = ops.build_pass_bwd(prec0, center, dirs, t, None, 0.0, False, packed, c2f, s0, o0, tuple(g), pose)
    ba.ws = w0.data_ptr()
    L.check(lib.sparf_launch_kernel(2, ctypes.byref(fa), ctypes.byref(ba), L.stream_ptr(d)), "wgrad")
    torch.cuda.synchronize()
    exact = float((gp2.double() - gp1.double()).norm() / gp2.double().norm())
    print(f"{base} R={R} N={N}: 8-bit weight-gradient kernel vs the plane kernel on the dequantised operands: {exact:.1e}")
    assert exact < 1e-6
    # ... and against the plane format's own gradients: the quantiser's error
    off, worst = 0, 0.0
    for (o, i) in L.LAYER_SHAPES:
        for n in (o * i, o):
            a, b = gp0[off:off + n].double(), gp1[off:off + n].double()
            worst = max(worst, float((a - b).norm() / a.norm().clamp_min(1e-30)))
            off += n
    total = float((gp0.double() - gp1.double()).norm() / gp0.double().norm())
    print(f"{base} R={R} N={N}: q8 vs plane weight gradient: worst tensor {worst:.2e}, all parameters {total:.2e}")
    assert worst < 3e-2 and total < 2e-2          # (small passes: the error of a sum over rows falls with the row count; DESIGN.md 6 has the benchmark shapes)


def test_q8_is_rejected_where_it_does_not_apply():
    lib = L.load()
    assert lib.sparf_save_bytes(L.PREC_FP32 | L.SAVE_Q8, 1024) == -1
    assert lib.sparf_bwd_workspace_bytes(L.PREC_FP32 | L.SAVE_Q8, 16, 64, 0) == -1
    opt, sd, center, dirs, t, g = _inputs(64, 32, 1)
    d = dev()
    plist = params_list(sd, d)
    prec = L.PREC_X3 | L.SAVE_Q8
    packed = ops.pack_weights(plist, prec)
    c2f = ops.c2f_weights(sd["progress"].to(d), opt.barf_c2f, d)
    with pytest.raises(L.SparfError):          # far rows are transplanted into plane saves only
        a, out, save, keep = ops.build_pass_fwd(prec, center, dirs, t, None, 0.0, False, packed, c2f, True, far=(4, L.PREC_FP32, ops.pack_weights(plist, L.PREC_FP32)))
        L.check(lib.sparf_pass_forward(ctypes.byref(a), L.stream_ptr(d)), "fwd")


@pytest.mark.parametrize("name", ["bf16+q8", "bf16x3+q8"])
def test_q8_training_step_through_the_graph(name):
    """Graph.render + fused loss + backward in an 8-bit-save mode against the plain mode: the same loss bit for bit, close gradients;
    then 40 Adam steps: the run trains"""
    from sparf_amd.optim import FusedAdam
    from tests.golden.recipe import ring_cameras
    from tests.test_graph_gpu import build_graph
    H, W, B = 8, 10, 2
    pose, intr = ring_cameras(B, H=H, W=W)
    rs = np.random.RandomState(4)
    target = torch.from_numpy(rs.uniform(size=(B, H * W, 3)).astype(np.float32)).to(dev())
    idx = torch.from_numpy(rs.permutation(H * W)[:32]).to(dev())
    res = {}
    for prec in (name[:-3], name):
        opt = small_opt(nerf=dict(rand_rays=64, sample_stratified=False, density_noise_reg=False), hip=dict(precision=prec))
        graph = build_graph(opt, 17)
        ret = graph.render(opt, pose.to(dev()), H=H, W=W, intr=intr.to(dev()), ray_idx=idx, depth_range=[1.2, 5.2], iter=0, mode="train")
        loss = ops.photometric_loss(ret.rgb, target[:, idx], rgb_fine=ret.rgb_fine)
        loss.backward()
        res[prec] = (float(loss), torch.cat([p.grad.reshape(-1) for p in graph.parameters() if p.grad is not None]).clone(), graph, opt)
    (l0, g0, _, _), (l1, g1, graph, opt) = res[name[:-3]], res[name]
    assert l0 == l1
    rel = float((g0.double() - g1.double()).norm() / g0.double().norm())
    print(f"{name}: gradient vs plane saves {rel:.2e}")
    assert rel < 1e-2
    adam = FusedAdam([graph.nerf, graph.nerf_fine], lr=1e-3)
    losses = []
    for it in range(40):
        adam.zero_grad(set_to_none=True)
        ii = torch.from_numpy(rs.permutation(H * W)[:32]).to(dev())
        ret = graph.render(opt, pose.to(dev()), H=H, W=W, intr=intr.to(dev()), ray_idx=ii, depth_range=[1.2, 5.2], iter=it, mode="train")
        loss = ops.photometric_loss(ret.rgb, target[:, ii], rgb_fine=ret.rgb_fine)
        loss.backward()
        adam.step()
        losses.append(float(loss.detach()))
    assert np.mean(losses[-5:]) < 0.6 * np.mean(losses[:5]), losses
