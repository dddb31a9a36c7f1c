"""Drop-in contract, checked on CPU: method signatures, state_dict layout, subclassing
pattern and option handling of sparf_amd's Graph / NeRF equal the reference's
(tests/golden/api.json was produced from /root/reference by tests/golden/make_golden.py)."""
import inspect
import json
import os

import pytest
import torch

from sparf_amd import lib as L
from sparf_amd.config import baseline_opt, default_opt
from sparf_amd.frequency_nerf import FrequencyEmbedder, NeRF, get_precision
from sparf_amd.renderer import Graph
from tests.golden.recipe import small_opt, make_state_dict

HERE = os.path.dirname(os.path.abspath(__file__))
API = json.load(open(os.path.join(HERE, "golden", "api.json")))


@pytest.mark.parametrize("cls", [Graph, NeRF, FrequencyEmbedder], ids=lambda c: c.__name__)
def test_signatures_match_reference(cls):
    ref = API[cls.__name__]
    for name, params in ref.items():
        if name in ("choose_activation", "compute_raw_density"):
            continue        # private helpers of the torch implementation, fused away here
        assert hasattr(cls, name), f"{cls.__name__}.{name} missing"
        sig = inspect.signature(getattr(cls, name))
        got = [[p.name, None if p.default is inspect._empty else repr(p.default)] for p in sig.parameters.values()]
        assert got == params, (name, got, params)


def test_state_dict_layout_and_strict_load():
    opt = small_opt()
    g = Graph(opt, torch.device("cpu"))
    sd = g.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == API["state_dict"]
    # a reference-style checkpoint dict loads with strict=True (base.py:219) and round-trips
    ck = {"nerf." + k: v for k, v in make_state_dict(opt, 1).items()}
    ck.update({"nerf_fine." + k: v for k, v in make_state_dict(opt, 2).items()})
    g.load_state_dict(ck, strict=True)
    assert torch.equal(g.nerf.mlp_feat[4].weight, ck["nerf.mlp_feat.4.weight"])
    assert "nerf.progress" in g.state_dict()
    assert sum(p.numel() for p in g.nerf.parameters()) == 530053


def test_trainer_patterns():
    opt = small_opt(barf_c2f=[0.4, 0.7])

    class Joint(Graph):                        # joint_pose_nerf_trainer.py:710-749
        def __init__(self, opt, device, pose_net):
            super().__init__(opt, device)
            self.pose_net = pose_net

        def get_w2c_pose(self, opt, data_dict, mode=None):
            return self.pose_net

    g = Joint(opt, torch.device("cpu"), torch.eye(3, 4)[None])
    assert g.get_network_components() == [g.nerf, g.nerf_fine]
    assert float(g.nerf.progress) == 0.0                # c2f starts at 0 (frequency_nerf.py:84)
    g.nerf.progress.data.fill_(0.3)                     # nerf_trainer.py:273-275
    g.re_initialize()
    assert float(g.nerf.mlp_feat[0].bias.abs().sum()) == 0.0
    opt_a = torch.optim.Adam([dict(params=g.nerf.parameters(), lr=1e-3)])
    opt_a.add_param_group(dict(params=g.nerf_fine.parameters(), lr=1e-3))      # nerf_trainer.py:181-185
    assert float(Graph(small_opt(), torch.device("cpu")).nerf.progress) == 1.0
    assert torch.allclose(g.get_c2w_pose(opt, None)[0, :, :3], torch.eye(3))


def test_no_cpu_fallback():
    opt = small_opt()
    g = Graph(opt, torch.device("cpu"))
    pose = torch.eye(3, 4)[None]
    intr = torch.tensor([[[7.0, 0, 4], [0, 7.0, 3], [0, 0, 1]]])
    with pytest.raises(L.SparfError):
        g.render(opt, pose, H=6, W=8, intr=intr, ray_idx=torch.arange(4), depth_range=[1.2, 5.2], mode="val")


def test_unsupported_architecture_is_loud():
    with pytest.raises(NotImplementedError):
        NeRF(default_opt(arch=dict(layers_feat=[None, 128, 128, 128])))


def test_default_precision_is_the_headline_mode(monkeypatch):
    """ONE default: what `Graph(opt, device)` of an unmodified trainer runs (nerf_trainer.py:112-114 passes
    no knob) is what bench.py measures by default -- bf16x3; inverse-depth passes route the last samples of every ray to fp32."""
    import bench
    from sparf_amd.frequency_nerf import DEFAULT_PRECISION, pass_precision, precision_name
    monkeypatch.delenv("SPARF_PRECISION", raising=False)
    monkeypatch.delenv("SPARF_INVERSE_DEPTH_PRECISION", raising=False)
    monkeypatch.delenv("SPARF_FAR_SAMPLES", raising=False)
    assert DEFAULT_PRECISION == "bf16x3" == bench.default_precision()
    o = default_opt()                                    # no opt.hip at all, as the reference's settings files
    assert precision_name(o) == "bf16x3"
    o.nerf.depth.param = "metric"
    assert pass_precision(o, 64) == (L.PREC_X3, None) == pass_precision(o, None)
    o.nerf.depth.param = "inverse"
    assert pass_precision(o, 64) == (L.PREC_X3, (8, L.PREC_FP32))          # Graph.render: last 8 samples of every ray in fp32
    assert pass_precision(o, 4) == (L.PREC_X3, (3, L.PREC_FP32))
    assert pass_precision(o, None) == (L.PREC_FP32, None) == pass_precision(o, 1)     # explicit points: whole pass
    assert pass_precision(o, None, to_max_samples=64) == (L.PREC_FP32, None)          # render_to_max WITH gradients: whole pass
    with torch.no_grad():                                                             # ... as its caller renders it: tiles by depth value
        assert pass_precision(o, None, to_max_samples=64) == (L.PREC_X3, (8.0, L.PREC_FP32))
        assert pass_precision(o, None, to_max_samples=48) == (L.PREC_FP32, None)      # not a multiple of the 32-row wave tile
    o.hip = dict(inverse_depth_precision="fp32")
    assert pass_precision(o, 64) == (L.PREC_FP32, None)
    o.hip = dict(inverse_depth_precision="bf16x3")
    assert pass_precision(o, 64) == (L.PREC_X3, None)
    o.hip = dict(far_samples=1)
    assert pass_precision(o, 64) == (L.PREC_X3, (1, L.PREC_FP32))
    monkeypatch.setenv("SPARF_PRECISION", "fp32")
    assert precision_name(default_opt()) == "fp32" and pass_precision(default_opt(), 64) == (L.PREC_FP32, None)


def test_q8_save_modes_change_the_saves_not_the_routing(monkeypatch):
    """'+q8' (8-bit save / gradient areas, C ABI 5) rides on the precision id of the passes that keep saves of their own; passes with
    far ROWS keep plane saves (the far rows are transplanted into bf16 planes), whole-fp32 passes are fp32 passes"""
    from sparf_amd.frequency_nerf import pass_precision
    for k in ("SPARF_PRECISION", "SPARF_INVERSE_DEPTH_PRECISION", "SPARF_FAR_SAMPLES"):
        monkeypatch.delenv(k, raising=False)
    assert L.base_prec(L.PREC_IDS["bf16x3+q8"]) == L.PREC_X3 and L.base_prec(L.PREC_IDS["bf16+q8"]) == L.PREC_BF16 and L.base_prec(L.PREC_FP32) == L.PREC_FP32
    o = default_opt()
    o.hip = dict(precision="bf16x3+q8")
    o.nerf.depth.param = "metric"
    assert pass_precision(o, 64) == (L.PREC_X3 | L.SAVE_Q8, None)
    o.nerf.depth.param = "inverse"
    assert pass_precision(o, 64) == (L.PREC_X3, (8, L.PREC_FP32))
    assert pass_precision(o, None) == (L.PREC_FP32, None)
    o.hip = dict(precision="bf16x3+q8", inverse_depth_precision="bf16x3")
    assert pass_precision(o, 64) == (L.PREC_X3 | L.SAVE_Q8, None)
    o.hip = dict(precision="bf16+q8")
    assert pass_precision(o, 64) == (L.PREC_BF16 | L.SAVE_Q8, None)
    o.hip = dict(precision="fp32+q8")
    with pytest.raises(ValueError):
        pass_precision(o, 64)


def test_options():
    assert get_precision(small_opt()) == L.PREC_FP32
    assert get_precision(small_opt(hip=dict(precision="bf16"))) == L.PREC_BF16
    o = baseline_opt(1)
    assert (o.nerf.rand_rays, o.nerf.sample_intvs, o.nerf.sample_intvs_fine, o.nerf.fine_sampling) == (4096, 64, 128, True)
    assert baseline_opt(2).barf_c2f == [0.4, 0.7]
