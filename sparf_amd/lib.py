"""ctypes binding of libsparf_hip.so (C ABI: include/sparf_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, the
product path raises.  The CPU oracle under `oracle/` is test infrastructure and is
never imported from here.
"""
import ctypes
import os
from ctypes import POINTER, c_float, c_int, c_int32, c_int64, c_void_p

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
# $SPARF_LIB selects another build of the same ABI (A/B kernel experiments)
LIB_PATH = os.environ.get("SPARF_LIB") or os.path.join(HERE, "libsparf_hip.so")

ABI_VERSION = 6                 # include/sparf_hip.h SPARF_ABI_VERSION
MAX_SEGMENTS = 16
PREC_BF16, PREC_FP32, PREC_X3 = 0, 1, 2
SAVE_Q8 = 16                    # include/sparf_hip.h SPARF_SAVE_Q8: OR-ed onto a pass's precision id = 8-bit save / gradient areas
PREC_IDS = {"bf16": PREC_BF16, "fp32": PREC_FP32, "bf16x3": PREC_X3, "bf16+q8": PREC_BF16 | SAVE_Q8, "bf16x3+q8": PREC_X3 | SAVE_Q8}


def base_prec(prec):
    """the plain precision id (arithmetic, packed weights, tables) of a pass precision that may carry SAVE_Q8"""
    return prec & ~SAVE_Q8
N_PARAMS = 530052
N_LAYERS = 10
# nn.Linear shapes in flat parameter order (W0,b0,...): (out, in)
LAYER_SHAPES = [(256, 63), (256, 256), (256, 256), (256, 256), (256, 319), (256, 256), (256, 256), (257, 256),
                (128, 283), (3, 128)]
PARAM_NAMES = [f"mlp_feat.{i}" for i in range(8)] + ["mlp_rgb.0", "mlp_rgb.1"]


class SparfError(RuntimeError):
    pass


class Segment(ctypes.Structure):
    _fields_ = [("ray0", c_int), ("nrays", c_int), ("noise_scale", c_float),
                ("g_rgb", c_void_p), ("g_depth", c_void_p), ("g_opacity", c_void_p), ("g_weights", c_void_p),
                ("g_depth_var", c_void_p), ("g_rgb_var", c_void_p), ("g_all_cumulated", c_void_p), ("g_density", c_void_p), ("g_rgb_samples", c_void_p)]


class PassFwd(ctypes.Structure):
    _fields_ = [("prec", c_int), ("nrays", c_int), ("nsamp", c_int),
                ("center", c_void_p), ("dir", c_void_p), ("t", c_void_p), ("noise", c_void_p),
                ("noise_scale", c_float), ("white_bg", c_int),
                ("packed", c_void_p), ("c2f", c_void_p), ("save", c_void_p), ("venc_ws", c_void_p),
                ("raylen", c_void_p), ("sigma_raw", c_void_p), ("rgb_samples", c_void_p), ("density", c_void_p),
                ("weights", c_void_p), ("rgb", c_void_p),
                ("depth", c_void_p), ("opacity", c_void_p), ("depth_var", c_void_p), ("rgb_var", c_void_p),
                ("all_cumulated", c_void_p), ("nseg", c_int), ("seg", POINTER(Segment)),
                ("far_count", c_int), ("far_prec", c_int), ("far_packed", c_void_p), ("far_ws", c_void_p), ("far_venc_ws", c_void_p),
                ("far_thr", c_float)]


class PassBwd(ctypes.Structure):
    _fields_ = [("prec", c_int), ("nrays", c_int), ("nsamp", c_int),
                ("center", c_void_p), ("dir", c_void_p), ("t", c_void_p), ("noise", c_void_p),
                ("noise_scale", c_float), ("white_bg", c_int),
                ("packed", c_void_p), ("c2f", c_void_p), ("tables", c_void_p), ("save", c_void_p),
                ("raylen", c_void_p), ("sigma_raw", c_void_p), ("rgb_samples", c_void_p), ("weights", c_void_p),
                ("g_rgb", c_void_p), ("g_depth", c_void_p), ("g_opacity", c_void_p), ("g_weights", c_void_p),
                ("ws", c_void_p), ("grad_params", c_void_p), ("d_center", c_void_p), ("d_dir", c_void_p),
                ("nseg", c_int), ("seg", POINTER(Segment)),
                ("g_depth_var", c_void_p), ("g_rgb_var", c_void_p), ("g_all_cumulated", c_void_p), ("g_density", c_void_p), ("g_rgb_samples", c_void_p),
                ("accumulate_rays", c_int)]


class CompositeFwd(ctypes.Structure):
    _fields_ = [("nrays", c_int), ("nsamp", c_int), ("white_bg", c_int),
                ("dir", c_void_p), ("t", c_void_p), ("density", c_void_p), ("rgb_samples", c_void_p),
                ("raylen", c_void_p), ("weights", c_void_p), ("rgb", c_void_p), ("depth", c_void_p), ("opacity", c_void_p),
                ("depth_var", c_void_p), ("rgb_var", c_void_p), ("all_cumulated", c_void_p)]


class CompositeBwd(ctypes.Structure):
    _fields_ = [("nrays", c_int), ("nsamp", c_int), ("white_bg", c_int),
                ("dir", c_void_p), ("t", c_void_p), ("density", c_void_p), ("rgb_samples", c_void_p), ("raylen", c_void_p), ("weights", c_void_p),
                ("g_rgb", c_void_p), ("g_depth", c_void_p), ("g_opacity", c_void_p), ("g_weights", c_void_p), ("g_depth_var", c_void_p),
                ("g_rgb_var", c_void_p), ("g_all_cumulated", c_void_p),
                ("d_density", c_void_p), ("d_rgb_samples", c_void_p), ("d_dir", c_void_p), ("d_len_ws", c_void_p)]


EXPORTS = {
    "sparf_abi_version": (c_int, []),
    "sparf_table_count": (c_int64, [c_int]),
    "sparf_build_tables": (c_int, [c_int, POINTER(c_int32)]),
    "sparf_stream_nchunks": (c_int, [c_int, c_int]),
    "sparf_stream_chunk": (c_int, [c_int, c_int, c_int, POINTER(c_int32)]),
    "sparf_packed_bytes": (c_int64, [c_int]),
    "sparf_pack_weights": (c_int, [c_int, POINTER(c_void_p), c_void_p, c_void_p, c_void_p]),
    "sparf_c2f_weights": (c_int, [c_void_p, c_int, c_float, c_float, c_void_p, c_void_p]),
    "sparf_ray_gen_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sparf_ray_gen_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p]),
    "sparf_adam_workspace_floats": (c_int64, []),
    "sparf_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_int,
                                c_float, c_void_p]),
    "sparf_adam_step_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_void_p,
                                    c_float, c_void_p]),
    "sparf_photometric_workspace_floats": (c_int64, []),
    "sparf_photometric_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sparf_sample_coarse": (c_int, [c_void_p, c_float, c_void_p, c_void_p, c_float, c_float, c_int, c_int, c_int, c_void_p, c_void_p]),
    "sparf_sample_fine": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sparf_sample_fine_hostgrid": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "sparf_save_bytes": (c_int64, [c_int, c_int64]),
    "sparf_pass_forward": (c_int, [POINTER(PassFwd), c_void_p]),
    "sparf_bwd_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "sparf_pass_backward": (c_int, [POINTER(PassBwd), c_void_p]),
    "sparf_composite_forward": (c_int, [POINTER(CompositeFwd), c_void_p]),
    "sparf_composite_backward": (c_int, [POINTER(CompositeBwd), c_void_p]),
    "sparf_launch_kernel": (c_int, [c_int, POINTER(PassFwd), POINTER(PassBwd), c_void_p]),
    "sparf_debug_wgrad_split": (c_int, [c_int64, c_int64, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "sparf_calib_mfma": (c_int64, [c_int, c_void_p, c_void_p]),
    "sparf_calib_hbm": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
}
CALIB_SINK_FLOATS = 1 << 18     # include/sparf_hip.h SPARF_CALIB_SINK_FLOATS

_lib = None
_tables_host = {}
_tables_dev = {}


def load():
    """Load the library (once).  Raises SparfError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SparfError(f"{LIB_PATH} not found: build it with `python -m sparf_amd.build` "
                             "(there is no CPU/PyTorch fallback for the renderer hot path)")
        lib = ctypes.CDLL(LIB_PATH)
        any_abi = bool(os.environ.get("SPARF_LIB") and os.environ.get("SPARF_ABI_ANY"))
        for name, (res, args) in EXPORTS.items():
            if any_abi and not hasattr(lib, name):
                continue                   # A/B run of an older kernel build: entry points added since are simply absent
            fn = getattr(lib, name)        # AttributeError if a declared symbol is missing
            fn.restype, fn.argtypes = res, args
        if lib.sparf_abi_version() != ABI_VERSION and not any_abi:
            raise SparfError("libsparf_hip.so ABI version mismatch")      # ($SPARF_ABI_ANY: A/B runs of an older kernel build, tools/ab_kernels.sh)
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise SparfError(f"{what} failed with code {rc}")


def ptr(t):
    if t is None:
        return None
    assert t.is_contiguous(), "sparf_hip needs dense tensors"
    return c_void_p(t.data_ptr())


def stream_ptr(device):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def on(device):
    """Device guard for the C-ABI calls: the library launches on the stream it is handed and sizes
    its grids from the CURRENT HIP device, so that device must be the tensors' device (torch ops
    do the same with their own guards); a Graph on cuda:1 while cuda:0 is current otherwise
    launches onto a foreign-device stream."""
    return torch.cuda.device(device)


def tables_host(prec):
    """Static gather tables (numpy int32), built by the library's host code."""
    if prec not in _tables_host:
        lib = load()
        n = lib.sparf_table_count(prec)
        if n <= 0:
            raise SparfError("bad precision")
        arr = np.empty(n, dtype=np.int32)
        check(lib.sparf_build_tables(prec, arr.ctypes.data_as(POINTER(c_int32))), "sparf_build_tables")
        _tables_host[prec] = arr
    return _tables_host[prec]


def tables_device(prec, device):
    key = (prec, str(device))
    if key not in _tables_dev:
        _tables_dev[key] = torch.from_numpy(tables_host(prec)).to(device)
    return _tables_dev[key]


_gpu_ok = False


def require_gpu(device):
    global _gpu_ok
    device = torch.device(device)
    if device.type == "cuda" and _gpu_ok:          # (checked once: torch.cuda.is_available() + load() cost ~5 us on every launch otherwise)
        return device
    if device.type != "cuda" or not torch.cuda.is_available():
        raise SparfError("the sparf_amd renderer runs on an MI355X (torch device 'cuda'); there is no CPU fallback")
    load()
    _gpu_ok = True
    return device
