"""The tape machinery of tests/callers_tape.py, checked without a GPU:
  * the committed tapes load, their draws regenerate from the numpy seed and match the taped checksums, the call mix and the
    BASELINE sizes are what tests/test_01_reference_tape_gpu.py expects (no reference needed: runs on the GPU box too);
  * where the reference tree is present (build container): a tape recorded from the reference and replayed ON THE REFERENCE reproduces
    every output exactly and the gradients to summation order -- so that on the GPU any difference is the HIP renderer's."""
import os

import numpy as np
import pytest
import torch

from tests import callers_tape as CT
from tests import ref_harness as RH

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["dtu_nerf", "dtu_barf", "llff_sparf", "replica_sparf"]


@pytest.mark.parametrize("name", NAMES)
def test_committed_tape_loads_and_regenerates_its_draws(name):
    tape = CT.load(os.path.join(ROOT, "tests", "golden", f"callers_tape_{name}.npz"))       # raises if a regenerated draw misses its checksum
    assert tape["rays"] == 4096 and tape["samples"] == (64, 128)
    kinds = [(c["method"], c["grad"]) for c in tape["calls"]]
    if name in ("dtu_nerf", "dtu_barf"):
        assert kinds == [("render", True)]
    else:
        assert kinds == [("render", True)] * 4 + [("render_to_max", False), ("render", True)]
        assert tape["calls"][5]["gpix"] is not None and tape["calls"][4]["depth_max"] is not None
    for c in tape["calls"]:
        n = c["pose"].shape[0] * (c["pixels"].shape[-2] if c["pixels"] is not None else c["ray_idx"].shape[-1])
        jit = [v for k, v in c["draws"] if k == ("rand", n * 64)]
        assert len(jit) == (1 if c["method"] == "render" else 0), "one stratified-jitter draw per render call (renderer.py:405-407)"
        for v in jit:
            assert float(v.min()) >= 0.0 and float(v.max()) < 1.0
        for k, g in c["gout"].items():
            assert g.shape == c["out"][k].shape and torch.isfinite(g).all()
    # the settings file with density noise carries the reference's merged fine depths (renderer.py:334-336) for the forced-depth replay
    for c in tape["calls"]:
        if name == "dtu_nerf":
            t = c["t_fine"]
            assert t is not None and tuple(t.shape) == (4, 1024, 192, 1) and bool((t[:, :, 1:] >= t[:, :, :-1]).all())
            rng = c["depth_range"][1] if c["depth_range"][0] == "tensor" else torch.tensor(c["depth_range"][1])
            assert float(t.min()) >= float(rng.min()) - 1e-4 and float(t.max()) <= float(rng.max()) + 1e-4
        else:
            assert c["t_fine"] is None
    # the opt document rebuilds into what Graph reads
    opt = CT.opt_from_json(tape["opt"], "fp32")
    assert opt.nerf.sample_intvs == 64 and opt.nerf.sample_intvs_fine == 128 and opt.nerf.fine_sampling and opt.hip.precision == "fp32"
    big = [n_ for n_, g in tape["grads"].items() if isinstance(g, dict)]
    assert len(big) == 2 * 7 and all(len(tape["grads"][n_]["idx"]) == CT.SUBSET for n_ in big)


def test_seeded_state_is_reproducible_and_reference_scaled():
    a, b = CT.seeded_state(1000), CT.seeded_state(1000)
    assert all(torch.equal(a[k], b[k]) for k in a)
    w = a["mlp_feat.1.weight"]
    assert abs(float(w.abs().max()) - np.sqrt(2.0) * np.sqrt(6.0 / 512)) < 1e-3          # xavier_uniform with relu gain (frequency_nerf.py:136-147)
    assert float(a["mlp_feat.7.weight"][0].abs().max()) <= np.sqrt(6.0 / 257) + 1e-6     # the density row: no gain


@pytest.mark.skipif(RH.reference_root() is None, reason="needs the reference tree (build container, or $SPARF_REFERENCE_ROOT)")
@pytest.mark.parametrize("name", ["dtu_barf", "llff_sparf"])
def test_tape_replays_exactly_on_the_reference_itself(name, tmp_path):
    tape = CT.record(name, seed=5, rays=256, samples=(8, 8), scene_hw=(60, 80))
    path = CT.save(tape, str(tmp_path / "tape.npz"))
    t2 = CT.load(path)
    opt = RH.load_settings(name, rays=256, samples=(8, 8), scene_hw=(60, 80))
    from source.models.renderer import Graph as RefGraph
    g = RefGraph(opt, "cpu")
    g.train()
    CT.load_seeded(g, t2["weight_seed"])
    r = CT.replay(t2, g, opt, "cpu")
    for e in r["per_call"]:
        assert all(v == 0.0 for k, v in e.items() if not k.startswith("_")), e
    assert r["grad_worst_tensor"] <= 1e-6 and not r["missing_grads"]
