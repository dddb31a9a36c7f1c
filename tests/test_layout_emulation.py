"""CPU emulation of the fused kernels' data flow, driven by the library's real tables.

The HIP kernels keep activations in MFMA register layout and consume weights from
permuted fragment streams.  This test replays exactly that index algebra in numpy
(float64) for single sample rows -- forward, dgrad, wgrad + un-permute -- using the
gather tables and chunk lists exported by libsparf_hip.so, and compares against the
oracle (torch autograd).  It needs no GPU: it proves the permutations, chunk order,
slot maps and buffer layouts are mutually consistent; the GPU tests then only have to
establish that the hardware MFMA lane maps are what layout.h says.
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O
from sparf_amd import lib as L
from tests.golden.recipe import small_opt, make_state_dict


# ---- python mirrors of layout.h (only the lane/register <-> index maps) -------------
def crow_of(q, h):
    return 32 * (q >> 4) + (q & 3) + 8 * ((q & 15) >> 2) + 4 * h


def qh_of_i(i):          # row i (0..31) inside an m-block -> (r, h)
    return (i & 3) | ((i >> 3) << 2), (i >> 2) & 1


def pos_of(q, h, ch):
    return (q // ch) * (2 * ch) + h * ch + (q % ch)


LAYER_OUT_MB = [8] * 7 + [9, 4, 1]
BIAS_OFF = np.concatenate([[0], np.cumsum([m * 32 for m in LAYER_OUT_MB])])


def chunks(prec, backward):
    lib = L.load()
    out = []
    buf = (ctypes.c_int32 * 8)()
    for i in range(lib.sparf_stream_nchunks(prec, backward)):
        assert lib.sparf_stream_chunk(prec, backward, i, buf) == 0
        out.append(dict(zip("layer seg mb0 nmb ks0 nks off bytes".split(), list(buf))))
    return out


class Emu:
    def __init__(self, prec, flat):
        self.prec = prec
        # bf16x3 shares the bf16 register layout; a logical stream element is a (head, tail) pair = 4 bytes
        self.KJ, self.CH, self.ab = {L.PREC_BF16: (8, 8, 2), L.PREC_FP32: (1, 4, 4), L.PREC_X3: (8, 8, 4)}[prec]
        t = L.tables_host(prec).astype(np.int64)
        lib = L.load()
        packed = lib.sparf_packed_bytes(prec)
        nbias = int(BIAS_OFF[-1])
        naux = nbias + 2 * 3 * 256                      # + raw-coordinate columns of layers 0 / 4 (streams.h xyz_pk)
        nstream = (packed - naux * 4) // self.ab        # blob = [fwd stream][bwd stream][packed bias][xyz_pk]
        tp = t[:nstream + naux]
        vals = np.where(tp >= 0, flat[np.clip(tp, 0, None)], 0.0)
        nf = sum(c["bytes"] for c in chunks(prec, 0)) // self.ab
        self.fwd, self.bwd = vals[:nf], vals[nf:nstream]
        self.bias = vals[nstream:nstream + nbias]
        self.xyz = vals[nstream + nbias:nstream + naux].reshape(2, 3, 8 * 32)      # [layer 0 | 4][coord][mb*32 + h*16 + r]
        # build option SP_XYZ_EXACT (streams.h): the forward stream then lacks the raw-coordinate columns of layers 0 / 4
        raw0 = np.arange(256)[:, None] * 63 + np.arange(3)[None]                    # W0[:, 0:3] in the flat parameter space
        self.xyz_exact = prec == L.PREC_X3 and not np.isin(raw0.reshape(-1), t[:nf]).any()
        self.wsrc = t[nstream + naux:]
        self.fch, self.bch = chunks(prec, 0), chunks(prec, 1)

    def frags(self, stream, c):
        n = c["nks"] * c["nmb"] * 64 * self.KJ
        o = c["off"] // self.ab
        return stream[o:o + n].reshape(c["nks"], c["nmb"], 64, self.KJ)

    def mma(self, stream, c, vec, D):
        """D[mb][i] += sum_k A[i][k] * B[k] for chunk c; vec[h][q] is the B operand."""
        f = self.frags(stream, c)
        for ks in range(c["nks"]):
            for m in range(c["nmb"]):
                for hk in range(2):
                    b = vec[hk][(c["ks0"] + ks) * self.KJ:(c["ks0"] + ks + 1) * self.KJ]
                    D[c["mb0"] + m] += f[ks, m, 32 * hk:32 * hk + 32, :] @ b

    # -------------------------------------------------------------- forward
    def forward(self, x0_ref, v_ref):
        """x0_ref [63], v_ref [27] in reference order -> dict of slot vectors + outputs."""
        x0 = np.zeros((2, 32))
        for h in range(2):                       # mirrors mlp_fwd.hip's encoding loop
            for i in range(15):
                arg = 15 * h + i
                coord = 2 if arg >= 20 else 1 if arg >= 10 else 0
                k = arg - 10 * coord
                x0[h, 2 * i] = x0_ref[3 + coord * 20 + k]
                x0[h, 2 * i + 1] = x0_ref[3 + coord * 20 + 10 + k]
        x0[0, 30], x0[0, 31], x0[1, 30] = x0_ref[0], x0_ref[1], x0_ref[2]
        v = np.zeros((2, 16))
        for h in range(2):                       # mirrors ray_setup_kernel (view_feat)
            for q in range(12):
                a = 6 * h + (q >> 1)
                v[h, q] = v_ref[3 + (a // 4) * 8 + (q & 1) * 4 + (a % 4)]
        v[0, 12], v[0, 13], v[1, 12] = v_ref[0], v_ref[1], v_ref[2]
        acts = {"x0": x0, "v": v}
        cur = x0
        res = {}
        for l in range(10):
            nmb = LAYER_OUT_MB[l]
            D = np.zeros((nmb, 32))
            for mb in range(nmb):
                for i in range(32):
                    r, h = qh_of_i(i)
                    D[mb, i] = self.bias[BIAS_OFF[l] + mb * 32 + h * 16 + r]
                    if self.xyz_exact and l in (0, 4):             # raw-coordinate columns as FMAs on the accumulator start
                        D[mb, i] += sum(self.xyz[0 if l == 0 else 1, c, mb * 32 + h * 16 + r] * x0_ref[c] for c in range(3))
            segs = {0: cur, 1: x0 if l == 4 else v}
            for c in [c for c in self.fch if c["layer"] == l]:
                self.mma(self.fwd, c, segs[c["seg"]], D)
            out = np.zeros((2, 16 * nmb))
            for mb in range(nmb):
                for i in range(32):
                    r, h = qh_of_i(i)
                    out[h, 16 * mb + r] = D[mb, i]
            if l == 7:
                res["sigma_raw"] = out[0, 128]
                out = out[:, :128]
            if l == 9:
                res["z"] = out[0, :3].copy()
            else:
                out = np.maximum(out, 0)
            acts[f"out{l}"] = out
            cur = out
        res["acts"] = acts
        return res

    # -------------------------------------------------------------- dgrad
    def backward(self, acts, dz, dsig):
        g = {}
        vec = np.zeros((2, 16)); vec[0, :3] = dz
        g["DZ"] = vec
        masks = {9: acts["out8"], 8: acts["out7"], 7: acts["out6"], 6: acts["out5"], 5: acts["out4"], 4: acts["out3"],
                 3: acts["out2"], 2: acts["out1"], 1: acts["out0"]}
        names = {9: "DG", 8: "DY7", 7: "DY6", 6: "DY5", 5: "DY4", 4: "DY3", 3: "DY2", 2: "DY1", 1: "DY0"}
        cur = vec
        dx0 = np.zeros((2, 32))
        for l in range(9, -1, -1):
            new_cur = None
            for seg in ((0, 1) if l in (4, 8) else (0,)):
                cs = [c for c in self.bch if c["layer"] == l and c["seg"] == seg]
                nmb = max(c["mb0"] + c["nmb"] for c in cs)
                D = np.zeros((nmb, 32))
                for c in cs:
                    self.mma(self.bwd, c, cur, D)
                out = np.zeros((2, 16 * nmb))
                for mb in range(nmb):
                    for i in range(32):
                        r, h = qh_of_i(i)
                        out[h, 16 * mb + r] = D[mb, i]
                if seg == 1 and l == 8:
                    g["dv"] = out
                elif (seg == 1 and l == 4) or l == 0:
                    dx0 += out
                else:
                    nxt = out * (masks[l] > 0)
                    if l == 8:          # append the raw-sigma slot (q = 128 on half 0)
                        nxt = np.concatenate([nxt, np.zeros((2, 16))], axis=1)
                        nxt[0, 128] = dsig
                    g[names[l]] = nxt
                    new_cur = nxt
            if new_cur is not None:
                cur = new_cur
        g["dx0"] = dx0
        return g

    # -------------------------------------------------------------- wgrad + un-permute
    def to_pos(self, vec):
        out = np.zeros(vec.shape[1] * 2)
        for h in range(2):
            for q in range(vec.shape[1]):
                out[pos_of(q, h, self.CH)] = vec[h, q]
        return out

    def param_grads(self, acts, g):
        P = self.to_pos
        save = {"XS": np.concatenate([P(acts["out3"]), P(acts["x0"])]), "H0": P(acts["out0"]), "H1": P(acts["out1"]),
                "H2": P(acts["out2"]), "H4": P(acts["out4"]), "H5": P(acts["out5"]), "H6": P(acts["out6"]),
                "FV": np.concatenate([P(acts["out7"]), P(acts["v"])]), "G": P(acts["out8"])}
        jobs = [("DY0", 8, "XS", 256, 2), ("DY1", 8, "H0", 0, 8), ("DY2", 8, "H1", 0, 8), ("DY3", 8, "H2", 0, 8),
                ("DY4", 8, "XS", 0, 10), ("DY5", 8, "H4", 0, 8), ("DY6", 8, "H5", 0, 8),
                ("DY7", 9, "H6", 0, 8), ("DG", 4, "FV", 0, 9), ("DZ", 1, "G", 0, 4)]
        mats, biases = [], []
        for gb, mb, sb, c0, nb in jobs:
            dy = P(g[gb])[:32 * mb]
            x = save[sb][c0:c0 + 32 * nb]
            mats.append(np.outer(dy, x).reshape(-1))
            biases.append(dy)
        partial = np.concatenate(mats + biases)
        return partial[self.wsrc]


def flat_params(sd):
    return np.concatenate([np.concatenate([sd[n + ".weight"].double().numpy().reshape(-1), sd[n + ".bias"].double().numpy()])
                           for n in L.PARAM_NAMES])


@pytest.mark.parametrize("prec", [L.PREC_BF16, L.PREC_FP32, L.PREC_X3], ids=["bf16", "fp32", "bf16x3"])
def test_emulated_kernels_match_oracle(prec):
    opt = small_opt()
    sd = {k: v.double() for k, v in make_state_dict(opt, 7).items()}
    for k, v in sd.items():
        if k != "progress":
            v.requires_grad_(True)
    rs = np.random.RandomState(3)
    pts = torch.from_numpy(rs.uniform(-1.5, 1.5, size=(1, 1, 1, 3)))
    ray = torch.from_numpy(rs.uniform(-1, 1, size=(1, 1, 3)))
    pts.requires_grad_(True)
    ray.requires_grad_(True)
    # oracle with explicit intermediate vectors
    x0 = torch.cat([pts, O.positional_encoding(opt, pts, 10, sd["progress"])], -1)
    d = torch.nn.functional.normalize(ray, dim=-1)[..., None, :]
    vv = torch.cat([d, O.positional_encoding(opt, d, 4, sd["progress"])], -1)
    x0.retain_grad(); vv.retain_grad()
    h = x0
    for li in range(8):
        if li == 4:
            h = torch.cat([h, x0], -1)
        h = torch.nn.functional.linear(h, sd[f"mlp_feat.{li}.weight"], sd[f"mlp_feat.{li}.bias"])
        if li == 7:
            raw, h = h[..., 0], h[..., 1:]
        h = torch.relu(h)
    g = torch.relu(torch.nn.functional.linear(torch.cat([h, vv], -1), sd["mlp_rgb.0.weight"], sd["mlp_rgb.0.bias"]))
    z = torch.nn.functional.linear(g, sd["mlp_rgb.1.weight"], sd["mlp_rgb.1.bias"])
    dz = rs.uniform(-1, 1, size=3)
    dsig = rs.uniform(-1, 1)
    loss = (z.reshape(-1) * torch.from_numpy(dz)).sum() + raw.reshape(-1)[0] * dsig
    loss.backward()

    flat = flat_params({k: v.detach() for k, v in sd.items()})
    emu = Emu(prec, flat)
    f = emu.forward(x0.detach().numpy().reshape(-1), vv.detach().numpy().reshape(-1))
    np.testing.assert_allclose(f["sigma_raw"], raw.item(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(f["z"], z.detach().numpy().reshape(-1), rtol=1e-10, atol=1e-12)

    gb = emu.backward(f["acts"], dz, dsig)
    # gradient w.r.t. the encoded point / view, mapped back to reference order
    dx0_ref = np.zeros(63)
    for hh in range(2):
        for i in range(15):
            arg = 15 * hh + i
            coord = 2 if arg >= 20 else 1 if arg >= 10 else 0
            k = arg - 10 * coord
            dx0_ref[3 + coord * 20 + k] = gb["dx0"][hh, 2 * i]
            dx0_ref[3 + coord * 20 + 10 + k] = gb["dx0"][hh, 2 * i + 1]
    dx0_ref[0], dx0_ref[1], dx0_ref[2] = gb["dx0"][0, 30], gb["dx0"][0, 31], gb["dx0"][1, 30]
    np.testing.assert_allclose(dx0_ref, x0.grad.numpy().reshape(-1), rtol=1e-9, atol=1e-12)
    dv_ref = np.zeros(27)
    for hh in range(2):
        for q in range(12):
            a = 6 * hh + (q >> 1)
            dv_ref[3 + (a // 4) * 8 + (q & 1) * 4 + (a % 4)] = gb["dv"][hh, q]
    dv_ref[0], dv_ref[1], dv_ref[2] = gb["dv"][0, 12], gb["dv"][0, 13], gb["dv"][1, 12]
    np.testing.assert_allclose(dv_ref, vv.grad.numpy().reshape(-1), rtol=1e-9, atol=1e-12)

    grads = emu.param_grads(f["acts"], gb)
    ref = np.concatenate([np.concatenate([sd[n + ".weight"].grad.numpy().reshape(-1), sd[n + ".bias"].grad.numpy()])
                          for n in L.PARAM_NAMES])
    np.testing.assert_allclose(grads, ref, rtol=1e-9, atol=1e-12)
