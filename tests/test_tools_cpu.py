"""CPU checks of the measurement tooling the committed evidence rests on (no GPU, no compute through the product path)."""
import importlib.util
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, path))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_mfma_gap_histogram_counts_instructions_between_mfmas():
    m = _load("tools/mfma_gap_hist.py", "mfma_gap_hist")
    asm = """
	.text
kernel:
	s_load_dword s0, s[0:1], 0x0
	v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]
	; a comment
	ds_read_b128 v[0:3], v8
	s_waitcnt lgkmcnt(0)
.LBB0_1:
	v_mfma_f32_32x32x16_bf16 a[0:15], v[0:3], v[4:7], a[0:15]
	v_mfma_f32_32x32x16_bf16 a[16:31], v[0:3], v[4:7], a[16:31]
	v_max_i32_e32 v0, 0, v0
	s_endpgm
"""
    assert m.gaps_of(asm) == [2, 0]          # instructions before the first and after the last MFMA do not count


def test_fp8_emulation_matches_torch_casts():
    """tests/tools/save_precision_study.py rounds with its own arithmetic (a per-tile scale, then the 8-bit grid): with the scale
    forced to one the grid must be torch's float8_e4m3fn / float8_e5m2 (round to nearest even, subnormals, saturation)."""
    m = _load("tests/tools/save_precision_study.py", "save_precision_study")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 32, generator=g) * torch.logspace(-4, 2.5, 64)[:, None]
    for dt, (mant, emin, vmax) in ((torch.float8_e4m3fn, (3, -6, 448.0)), (torch.float8_e5m2, (2, -14, 57344.0))):
        xs = x / x.abs().max() * vmax        # amax = vmax -> scale 2^0 (one tile: all rows); magnitudes down to the subnormals
        got = m.q_fp8_tiles(xs, mant, emin, vmax, tile=64)
        ref = xs.to(dt).float()
        assert torch.equal(got, ref)
    # scaled: relative error of an e4m3 grid is <= 2^-4 for values within 2^-6 of the tile's largest
    q = m.q_fp8_tiles(x, 3, -6, 448.0)
    big = x.abs() >= x.abs().reshape(2, -1).amax(dim=1).repeat_interleave(32)[:, None] * 2.0 ** -6
    assert float(((q - x).abs() / x.abs())[big].max()) <= 2.0 ** -4 + 1e-6


def test_emulated_linear_is_the_plain_one_without_rounding():
    m = _load("tests/tools/save_precision_study.py", "save_precision_study2")
    import types
    cfg = types.SimpleNamespace(fwd="fp32", dy="fp32", dgrad_w="fp32", save_x="fp32", save_dy="fp32")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 5, 8, generator=g, requires_grad=True)
    w = torch.randn(4, 8, generator=g, requires_grad=True)
    b = torch.randn(4, generator=g, requires_grad=True)
    up = torch.randn(3, 5, 4, generator=g)
    got = torch.autograd.grad((m.EmuLinear.apply(x, w, b, cfg) * up).sum(), (x, w, b))
    ref = torch.autograd.grad((torch.nn.functional.linear(x, w, b) * up).sum(), (x, w, b))
    for a, r in zip(got, ref):
        assert torch.allclose(a, r, rtol=1e-5, atol=1e-6)


def test_shipped_kernels_have_no_experiment_or_probe_flag_on():
    """The kernel sources carry measured experiments as build flags: the defaults are the shipped configuration,
    the probes give WRONG RESULTS by design and must never be defined in the sources or by sparf_amd.build."""
    import re
    from sparf_amd import build as B
    src = {f: open(os.path.join(B.CSRC, f)).read() for f in os.listdir(B.CSRC) if f.endswith((".h", ".hip", ".cpp"))}
    text = "\n".join(src.values())
    expected = {"SP_BWD_DEFER": "1", "SP_DEFER_EPI": "1", "SP_BWD_SPREAD": "1", "SP_BWD_STAGGER": "0", "SP_X3_DGRAD_WAVES": "0", "SP_X3_DGRAD_PARTS": "2",
                "SP_WG_SPREAD": "0", "SP_WG_Q8_HALVES": "0", "SP_SAVE_AUX": "2", "SP_XYZ_EXACT": "0", "SP_LAZY_ACC_READ": "1", "SP_SLOT_BALANCE": "1"}
    for name, val in expected.items():
        m = re.search(r"#ifndef %s\s*\n#define %s (\S+)" % (name, name), text)
        assert m, f"{name}: no guarded default found"
        assert m.group(1) == val, (name, m.group(1), "shipped default is", val)
    for name in ("SP_PROBE_NO_STORES", "SP_PROBE_NO_DMA", "SP_PROBE_NO_BARRIER", "SP_PROBE_HALF_SAVES", "SP_PROBE_NO_ENCODING", "SP_PROBE_NO_TILE_END", "SP_PROF",
                 "SP_X3_DGRAD_FULL"):
        assert not re.search(r"^\s*#\s*define\s+%s\b" % name, text, flags=re.M), f"{name} is defined in the sources"
        assert not any(name in f for f in B.FLAGS), f"{name} is passed by the default build"
