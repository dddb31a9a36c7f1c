"""A whole training step (ray generation -> both passes -> loss -> backward -> clip + Adam) captured in ONE hipGraph
through the C ABI and replayed (bench_workloads.Workload.capture).  What must hold for that to be legal:
no pass call touches the host side of HIP (cached CU count), all randomness is device-side, and Adam's step count
lives on the device (sparf_adam_step_dev) -- a host-side count would freeze at its capture value.
Run with `pytest -m gpu`."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("config", [1, 2])
def test_training_step_replays_as_one_hipgraph(config):
    from bench_workloads import Workload
    dev = torch.device("cuda:0")
    w = Workload(config, "bf16x3", dev, rays=510 if config == 2 else 512, graph_capture=True, seed=3)
    step = w.capture(warmup=2)
    params = [p for p in w.net_params]
    before = [p.detach().clone() for p in params]
    state = w.optim.state[w.optim.param_groups[0]["params"][0]]
    n0 = int(state["step_dev"])                      # the warm-up steps ran; the captured one only recorded
    assert n0 == 2
    losses = [float(step()) for _ in range(24)]
    torch.cuda.synchronize()
    assert all(math.isfinite(x) for x in losses)
    assert int(state["step_dev"]) == n0 + 24         # the device-side count advanced once per replay
    assert any(not torch.equal(a, b) for a, b in zip(before, params))
    assert sum(losses[-6:]) < sum(losses[:6])         # and the replays train: Adam's bias corrections follow the device count
    # different rays and draws every replay (ray indices drawn outside the graph, jitter / noise from the graph-safe device RNG)
    assert len({round(x, 7) for x in losses}) > 20
    # an eager step of the same workload still works next to the captured graph
    eager = float(w._step_with(torch.randperm(w.H * w.W, device=dev)[:w.rays // w.B]))
    assert math.isfinite(eager)
