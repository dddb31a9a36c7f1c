"""Parity at the benchmark's own shapes: one full Graph render + backward per
BASELINE.json config 1-4, in the fp32 parity mode and in the bf16x3 headline mode, outputs AND every
parameter / ray / pose gradient against the oracle's float64 referee (tests/scale_cases.py; the
referee's PyTorch code runs on the GPU in float64 here -- tests/tools/scale_parity.py --referee-device cpu
gave the same numbers on the CPU, profiles/r02b_parity_scale.json).

Gradient bounds = 1.5 x the measured, deterministic values of round 6 (profiles/r06_test00_measured.json, written by this file; VERDICT r05
next-6: rounds 2-5 kept 2-4 x head-room).

Outputs, max|a-b| / max|b| per tensor: <= 1e-4 (north_star) for every key in both modes at the
metric-depth configs 1, 2, 4 (measured fp32 2e-6, bf16x3 2.8e-5).  Config 3 samples inverse depth
(renderer.py:413-416): t reaches 1e8 and the network is evaluated at |p| ~ 1e8, where the fp32
REFERENCE's own per-sample values sit 5e-5 .. 2e-4 from the referee; what the losses read (rgb, depth,
opacity, weights, depth_var) is held to 1e-4 in fp32 (measured 1.5e-5) and in bf16x3 with its far rows in fp32, 3e-4 in bf16x3 without (1.1e-4),
the per-sample values and all_cumulated -- returned but never consumed from `render` (SURVEY 8
quirk 12) -- to 2e-3 / 5e-2.

Gradients, relative L2 per tensor.  The floor is not rounding but ReLU decisions: a forward error e
flips the units whose pre-activation is within e of zero, and a gradient sum over 786 k rows with
incoherent signs (this test's random loss functional is the worst case) sees that as a relative
error that does not average out.  Measured worst tensor (always mlp_feat.0.weight, shrinking towards
the output layers): fp32 reference 1.0e-3 .. 1.5e-3, HIP fp32 0.8e-3 .. 1.7e-3, bf16x3 6e-3 .. 8e-3 --
and 5.2e-3 with the full-precision backward (-DSP_X3_SAVE_PLANES=2 -DSP_X3_DGRAD_FULL: rgb-layer
gradient 6e-6, feature layers unchanged), i.e. the bf16-rounded backward operands of the default
build add ~30 % to a floor set by the forward's 2e-5; DESIGN.md section 2 has the table.
Run with `pytest -m gpu`."""
import json
import os

import pytest

from tests import scale_cases as S

pytestmark = pytest.mark.gpu

_MEASURED = {}


def _note(name, e, extra=None):
    """what this run measured, next to the bounds it is held to (gpurun_out/r06_test00_measured.json -> profiles/): the bounds below are
    1.5 x these deterministic values"""
    keys = ("rendered_worst", "per_sample_worst", "param_grad_rel_l2_worst", "param_grad_rel_l2_all", "d_origins_rel_l2", "d_viewdirs_rel_l2", "d_pose_maxrel")
    _MEASURED[name] = dict({k: e[k] for k in keys if k in e}, **(extra or {}))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "r06_test00_measured.json"), "w") as f:
        json.dump(_MEASURED, f, indent=1)

OUT_TOL = 1e-4
#                                                    measured maximum over configs 1-4 and the variants (fp32 / bf16x3)
GRAD_WORST = {"fp32": 2.4e-3, "bf16x3": 1.05e-2}   # worst parameter tensor, relative L2:                       1.57e-3 / 7.00e-3
GRAD_ALL = {"fp32": 8e-4, "bf16x3": 3.7e-3}        # all parameters of both networks as one vector:             5.25e-4 / 2.42e-3
RAYGRAD = {"fp32": 2.3e-3, "bf16x3": 1.04e-2}      # d origins, d viewdirs:                                     1.53e-3 / 6.93e-3
POSEGRAD = {"fp32": 2.3e-3, "bf16x3": 2.1e-2}      # max-norm relative, through the float64 ray generation:     1.48e-3 / 1.38e-2 (config 3, no far rows)
CONFIG0_WORST = {"fp32": 4.2e-4, "bf16x3": 1.21e-2}    # 255 rays, 65 k rows: a single flipped ReLU weighs more:  2.74e-4 / 8.06e-3
# inverse depth (config 3): the bf16x3 mode routes the last 8 samples of every ray through the fp32 kernels (frequency_nerf.
# pass_precision, C ABI "far rows"; profiles/r04_inverse_routing_study.json), so both modes are held to 1e-4 on what the losses
# read; "bf16x3#" = whole passes on the fp32 kernels (round 3), "bf16x3!" = no correction (opt.hip.inverse_depth_precision)
INVERSE_RENDERED = {"fp32": 1e-4, "bf16x3": 1e-4, "bf16x3#": 1e-4, "bf16x3!": 3e-4}
INVERSE_PER_SAMPLE = {"fp32": 2e-3, "bf16x3": 2e-3, "bf16x3#": 2e-3, "bf16x3!": 5e-2}


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "bf16x3#", "bf16x3!"])
@pytest.mark.parametrize("cfg", [1, 2, 3, 4])
def test_benchmark_shape_parity(cfg, precision):
    if precision[-1] in "!#" and cfg != 3:
        pytest.skip("the inverse-depth variants only exist for inverse depth")
    r = S.run_case(cfg, precision, referee_device="cuda:0", chunk=1024)
    e = r["hip"]
    print(json.dumps({k: v for k, v in r.items() if k != "hip"}))
    print(json.dumps({k: v for k, v in e.items() if k != "param_grad_rel_l2"}))
    _note(f"config{cfg}/{precision}", e, dict(to_max=r.get("to_max")))
    assert r["t_coarse_bit_exact"] and r["t_fine_sorted"]
    # resampled depths: a few fp32 ulps of the bin range (pdf division / cdf rounding)
    assert r["t_fine_vs_sampler_oracle_maxabs"] <= 2e-5 * max(abs(S.CONFIGS[cfg]["rng"][0]), abs(S.CONFIGS[cfg]["rng"][1]), 1.0)
    if cfg == 3:
        assert e["rendered_worst"] <= INVERSE_RENDERED[precision], e["outputs"]
        assert e["per_sample_worst"] <= INVERSE_PER_SAMPLE[precision], e["outputs"]
    else:
        bad = {k: v for k, v in e["outputs"].items() if not v <= OUT_TOL}
        assert not bad, bad
    precision = precision.rstrip("!#")
    assert e["param_grad_rel_l2_worst"] <= GRAD_WORST[precision], {k: v for k, v in e["param_grad_rel_l2"].items() if v > GRAD_WORST[precision]}
    assert e["param_grad_rel_l2_all"] <= GRAD_ALL[precision]
    if "d_origins_rel_l2" in e:
        assert e["d_origins_rel_l2"] <= RAYGRAD[precision] and e["d_viewdirs_rel_l2"] <= RAYGRAD[precision]
        assert e["d_pose_maxrel"] <= POSEGRAD[precision]
    if "to_max" in r:
        assert r["to_max_t_bit_exact"]
        assert max(r["to_max"].values()) <= OUT_TOL, r["to_max"]


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_reference_default_sample_counts(precision):
    """The reference's own defaults, which BASELINE overrides: 128 coarse + 128 fine samples
    (default_config.py:114,117), 2048 rays (:118) -- 256 + 128 rows per ray, other tile counts, same bounds."""
    r = S.run_case(1, precision, referee_device="cuda:0", chunk=512, nc=128, nf=128, rays_scale=0.5)
    e = r["hip"]
    print(json.dumps({k: v for k, v in e.items() if k != "param_grad_rel_l2"}))
    _note(f"defaults128+128/{precision}", e)
    assert r["t_coarse_bit_exact"] and r["t_fine_sorted"] and r["rays"] == 2048
    bad = {k: v for k, v in e["outputs"].items() if not v <= OUT_TOL}
    assert not bad, bad
    assert e["param_grad_rel_l2_worst"] <= GRAD_WORST[precision] and e["param_grad_rel_l2_all"] <= GRAD_ALL[precision]


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_config0_shape(precision):
    """BASELINE configs[0]: the CPU-runnable case, 3 views x 85 rays x (64 + 128) -- 255 rays, ragged against every
    tile size (32-row waves, 128 / 256-row workgroups, wgrad split ranges)."""
    r = S.run_case(2, precision, referee_device="cuda:0", chunk=512, rays_scale=85 / 1365)
    e = r["hip"]
    _note(f"config0/{precision}", e)
    assert r["rays"] == 255 and r["t_coarse_bit_exact"]
    bad = {k: v for k, v in e["outputs"].items() if not v <= OUT_TOL}
    assert not bad, bad
    # 65 k rows instead of 786 k: a single flipped ReLU weighs more, same floor mechanism
    assert e["param_grad_rel_l2_worst"] <= CONFIG0_WORST[precision]
