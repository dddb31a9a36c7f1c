"""Host logic of the native library, no GPU needed: the C ABI loads and exports every
symbol of include/sparf_hip.h, and the static permutation tables are bijective where
they must be (every weight exactly once per stream, every parameter gets a gradient
source)."""
import ctypes
import os
import re

import numpy as np
import pytest

from sparf_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    hdr = open(os.path.join(ROOT, "include", "sparf_hip.h")).read()
    declared = set(re.findall(r"\b(sparf_[a-z0-9_]+)\s*\(", hdr))
    declared = {d for d in declared if not d.endswith("_t")}
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.sparf_abi_version() == L.ABI_VERSION == int(re.search(r"#define SPARF_ABI_VERSION (\d+)", hdr).group(1))


def _chunk_bytes(lib, prec, backward):
    buf = (ctypes.c_int32 * 8)()
    out = []
    for i in range(lib.sparf_stream_nchunks(prec, backward)):
        assert lib.sparf_stream_chunk(prec, backward, i, buf) == 0
        out.append(buf[7])
    return out


def param_layout():
    offs, o = [], 0
    for (out, inp) in L.LAYER_SHAPES:
        offs.append((o, o + out * inp, o + out * inp + out))
        o += out * inp + out
    assert o == L.N_PARAMS
    return offs


@pytest.mark.parametrize("prec", [L.PREC_BF16, L.PREC_FP32, L.PREC_X3])
def test_tables(prec):
    lib = L.load()
    t = L.tables_host(prec)
    n_w = sum(o * i for o, i in L.LAYER_SHAPES)
    ab = 2 if prec == L.PREC_BF16 else 4          # bytes per logical stream element (bf16x3: head + tail)
    packed = lib.sparf_packed_bytes(prec)
    n_bias = (7 * 8 + 9 + 4 + 1) * 32
    n_aux = n_bias + 2 * 3 * 256                  # + the raw-coordinate columns of layers 0 / 4 (streams.h xyz_pk)
    n_stream = (packed - n_aux * 4) // ab
    assert len(t) == n_stream + n_aux + L.N_PARAMS
    is_weight = np.zeros(L.N_PARAMS, bool)
    for w0, w1, b1 in param_layout():
        is_weight[w0:w1] = True
    # raw-coordinate columns: W0[:, 0:3] and W4[:, 256:259], each exactly once in the xyz table
    lay = param_layout()
    raw = np.concatenate([lay[0][0] + np.arange(256)[:, None] * 63 + np.arange(3)[None],
                          lay[4][0] + np.arange(256)[:, None] * 319 + 256 + np.arange(3)[None]]).reshape(-1)
    xyz = t[n_stream + n_bias:n_stream + n_aux]
    assert sorted(xyz.tolist()) == sorted(raw.tolist())
    is_raw = np.zeros(L.N_PARAMS, bool)
    is_raw[raw] = True
    streams = t[:n_stream]
    used = streams[streams >= 0]
    assert is_weight[used].all(), "weight streams must not reference biases"
    # forward and backward stream each contain every weight exactly once -- except that a bf16x3 FORWARD stream built with
    # -DSP_XYZ_EXACT=1 leaves the raw-coordinate columns to the fp32 FMAs on the accumulator start (they stay in the dgrad stream)
    counts = np.bincount(used, minlength=L.N_PARAMS)
    nf = sum(_chunk_bytes(lib, prec, 0)) // ab
    xyz_exact = prec == L.PREC_X3 and not is_raw[streams[:nf][streams[:nf] >= 0]].any()      # build option SP_XYZ_EXACT (streams.h)
    if xyz_exact:
        assert (counts[is_weight & ~is_raw] == 2).all() and (counts[is_raw] == 1).all()
        keep = (streams >= 0) & ~is_raw[np.clip(streams, 0, None)]
    else:
        assert (counts[is_weight] == 2).all()
        keep = streams >= 0
    # split point: the first half (forward) alone is a bijection too
    pos = np.flatnonzero(keep)
    order = np.argsort(streams[pos], kind="stable")
    sorted_idx, sorted_pos = streams[pos][order], pos[order]
    firsts, seconds = sorted_pos[0::2], sorted_pos[1::2]
    assert (sorted_idx[0::2] == sorted_idx[1::2]).all()
    # the forward stream precedes the backward stream in the table
    assert firsts.max() < seconds.min()
    # biases: each exactly once in the packed-bias table
    bias = t[n_stream:n_stream + n_bias]
    bused = bias[bias >= 0]
    assert (~is_weight[bused]).all() and len(np.unique(bused)) == len(bused) == (~is_weight).sum()
    # wgrad source: every parameter has one, all distinct
    wsrc = t[n_stream + n_aux:]
    assert (wsrc >= 0).all() and len(np.unique(wsrc)) == L.N_PARAMS


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.SparfError):
        L.load()


def test_c_abi_rejects_bad_arguments_without_a_gpu():
    """Error convention of the C ABI (include/sparf_hip.h): non-zero return codes, never a
    crash, for NULL structs / bad precisions / missing buffers -- all decided before any HIP
    call, so this runs without a GPU."""
    lib = L.load()
    assert lib.sparf_abi_version() == L.ABI_VERSION
    assert lib.sparf_table_count(7) == -1 and lib.sparf_packed_bytes(-1) == -1 and lib.sparf_save_bytes(9, 100) == -1
    assert lib.sparf_build_tables(0, None) != 0
    assert lib.sparf_stream_nchunks(5, 0) == -1
    assert lib.sparf_pass_forward(None, None) != 0 and lib.sparf_pass_backward(None, None) != 0
    fwd = L.PassFwd(prec=0, nrays=4, nsamp=8)              # all pointers NULL
    assert lib.sparf_pass_forward(ctypes.byref(fwd), None) != 0
    fwd = L.PassFwd(prec=3, nrays=4, nsamp=8)
    assert lib.sparf_pass_forward(ctypes.byref(fwd), None) != 0
    fwd = L.PassFwd(prec=0, nrays=0, nsamp=8)              # empty batch: nothing to do, ok
    assert lib.sparf_pass_forward(ctypes.byref(fwd), None) == 0
    assert lib.sparf_sample_coarse(None, 0.5, None, None, 1.0, 2.0, 0, 16, 0, None, None) != 0
    assert lib.sparf_sample_coarse(None, 0.5, None, None, 1.0, 2.0, 0, 16, 8, None, None) != 0     # no output buffer
    assert lib.sparf_sample_fine(None, None, None, None, 1.0, 2.0, 4, 8, 8, None, None, None) != 0
    assert lib.sparf_ray_gen_forward(None, None, None, None, 0, 4, 1, 3, None, None, None) != 0
    assert lib.sparf_adam_step(None, None, None, None, None, None, 1e-3, 0.9, 0.999, 1e-8, 1, 0.0, None) != 0
    assert lib.sparf_photometric_loss(None, None, None, 10, 0, 0.5, None, None, None, None, None) != 0
    assert lib.sparf_photometric_workspace_floats() >= 1
    assert lib.sparf_c2f_weights(None, 1, 0.4, 0.7, None, None) != 0            # no output
    out16 = ctypes.c_void_p(16)
    assert lib.sparf_c2f_weights(None, 1, 0.4, 0.7, out16, None) != 0           # c2f on but no progress pointer
    assert lib.sparf_c2f_weights(out16, 1, 0.4, 0.4, out16, None) != 0          # empty band range
    assert lib.sparf_pack_weights(0, None, None, None, None) != 0
    assert lib.sparf_bwd_workspace_bytes(0, 4096, 192, 0) > 0 and lib.sparf_bwd_workspace_bytes(0, -1, 192, 0) == -1
    # sizes scale with the precision's bytes per saved element (bf16x3 saves the bf16 head plane)
    assert lib.sparf_save_bytes(1, 4096) > lib.sparf_save_bytes(0, 4096) and lib.sparf_save_bytes(2, 4096) == lib.sparf_save_bytes(0, 4096)


def test_wgrad_split_of_an_active_range_fits_the_workspace():
    """the split-K count of the weight gradient is not monotone in the row count, so the split of an
    ACTIVE sub-range of a segmented pass could exceed the number of partial blocks the workspace was sized for (12289 rays x
    64 samples -> 127 splits, an active range of 8684 rays -> 128: one 2 MB block written past `partial`).  The capped split
    never does, still covers the range, and keeps 64-row-aligned splits."""
    import ctypes
    import numpy as np
    lib = L.load()
    nt, na, rps = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()

    def split(total, active):
        assert lib.sparf_debug_wgrad_split(total, active, ctypes.byref(nt), ctypes.byref(na), ctypes.byref(rps)) == 0
        return nt.value, na.value, rps.value

    t, a, r = split(12289 * 64, 8684 * 64)              # the advisor's counter-example
    assert t == 127 and a <= t and a * r >= 8684 * 64
    rs = np.random.RandomState(0)
    cases = [(int(tr) * ns, int(ar) * ns) for ns in (64, 192) for tr in rs.randint(1, 40000, size=400) for ar in rs.randint(1, tr + 1, size=8)]
    cases += [(n, n) for n in (0, 1, 63, 64, 65, 4096, 524288, 786432, 1 << 27)] + [(786432, 0), (786432, 32)]
    for total, active in cases:
        t, a, r = split(total, active)
        assert 1 <= a <= t <= 128, (total, active, t, a)
        assert r % 64 == 0 and r >= 64 and a * r >= active, (total, active, a, r)
        assert (a - 1) * r < max(active, 1), (total, active, a, r)      # no empty trailing split
