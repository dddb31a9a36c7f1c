"""Ray-batch data parallelism with the HIP path inside each rank: two ranks share
the one GPU of the test box (gloo carries the exchange; RCCL needs one device per rank), each
renders its shard of a global ray set through `Graph`, and the exchanged gradients -- both
networks' flat buffers in place, pose parameters and the loss / NaN scalars in a small bucket,
mirroring iter_based_trainer.py:248-252 -- must equal the single-rank step on the whole set.
Also: `python bench.py --gpus 2` started plainly re-launches itself under torch.distributed.run.
Run with `pytest -m gpu`."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    from tests.golden.recipe import make_state_dict, ring_cameras, small_opt
    opt = small_opt(barf_c2f=[0.1, 0.5], nerf=dict(rand_rays=64, sample_stratified=False))
    H, W, B, N = 12, 16, 2, 42
    pose, intr = ring_cameras(B, H=H, W=W)
    rs = np.random.RandomState(5)
    idx = torch.from_numpy(rs.permutation(H * W)[:N])
    target = torch.from_numpy(rs.uniform(size=(B, N, 3)).astype(np.float32))
    return opt, make_state_dict, H, W, B, N, pose, intr, idx, target


def _step(lo, hi, seed_shift=0):
    """gradients of the global-mean photometric loss restricted to rays [lo, hi) of the global set"""
    from bench_workloads import PoseGraph
    opt, make_sd, H, W, B, N, pose, intr, idx, target = _problem()
    dev = torch.device("cuda:0")
    graph = PoseGraph(opt, dev, pose)
    graph.nerf.load_state_dict(make_sd(opt, 90 + seed_shift, 0.3))
    graph.nerf_fine.load_state_dict(make_sd(opt, 91 + seed_shift, 0.3))
    with torch.no_grad():
        graph.se3_refine.copy_(torch.tensor([[0.02, -0.01, 0.015, 0.05, -0.03, 0.02], [-0.01, 0.02, 0.0, -0.02, 0.04, 0.01]], device=dev))
    return graph, opt, (H, W, B, N, intr, idx, target, dev), (lo, hi)


def _loss(graph, opt, ctx, span):
    H, W, B, N, intr, idx, target, dev = ctx
    lo, hi = span
    p = graph.get_w2c_pose(opt, None)
    ret = graph.render(opt, p, H=H, W=W, intr=intr.to(dev), ray_idx=idx[lo:hi].to(dev), depth_range=[1.2, 5.2], iter=10, mode="train")
    tgt = target[:, lo:hi].to(dev)
    return (((ret.rgb - tgt) ** 2).sum() + ((ret.rgb_fine - tgt) ** 2).sum()) / (B * N * 3)      # global normaliser


def _worker(rank, world, port, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sparf_amd.parallel import GradBucket, broadcast_parameters, shard_slice
    N = _problem()[5]
    lo, hi = shard_slice(N, rank, world)
    graph, opt, ctx, span = _step(lo, hi, seed_shift=10 * rank)      # different weights per rank ...
    broadcast_parameters(graph, src=0)                                # ... made identical here (and the packed-weight cache told)
    loss = _loss(graph, opt, ctx, span)
    loss.backward()
    # ONE exchange per step: both networks' flat gradient buffers + the pose gradient + the scalars in a single message
    bucket = GradBucket([p for net in (graph.nerf, graph.nerf_fine) for n, p in net.named_parameters() if n != "progress"] + [graph.se3_refine])
    extra = bucket.allreduce_(average=False, extra=torch.stack([loss.detach(), torch.isnan(loss.detach()).float(),
                                                                torch.tensor(float(hi - lo), device=loss.device)]))
    assert bucket.last_path == "flat" and bucket.collectives == 1     # the HIP backward's flat buffers: a cat of three pieces, one all-reduce
    if rank == 0:
        out = {f"{n}.{k}": p.grad.cpu().numpy().copy() for n, net in (("nerf", graph.nerf), ("nerf_fine", graph.nerf_fine))
               for k, p in net.named_parameters() if k != "progress"}
        out["se3"] = graph.se3_refine.grad.cpu().numpy().copy()
        q.put((out, extra.cpu().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_hip_step_equals_single_rank():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    grads, extra = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    N = _problem()[5]
    graph, opt, c, span = _step(0, N)
    loss = _loss(graph, opt, c, span)
    loss.backward()
    assert abs(float(extra[0]) - float(loss.detach())) < 2e-6 * abs(float(loss.detach())) and float(extra[1]) == 0.0 and float(extra[2]) == N
    worst = 0.0
    for n, net in (("nerf", graph.nerf), ("nerf_fine", graph.nerf_fine)):
        for k, p in net.named_parameters():
            if k == "progress":
                continue
            ref = p.grad.cpu().numpy()
            worst = max(worst, float(np.abs(grads[f"{n}.{k}"] - ref).max() / (np.abs(ref).max() + 1e-30)))
    assert worst < 2e-5, worst                                        # same rows, different split-K grouping of the row sum
    ref = graph.se3_refine.grad.cpu().numpy()
    assert np.abs(grads["se3"] - ref).max() < 1e-4 * np.abs(ref).max()


def _nccl_worker(rank, world, port, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from sparf_amd.parallel import GradBucket, broadcast_parameters, shard_slice
    from bench_workloads import Workload
    w = Workload(1, "bf16x3", torch.device("cuda", rank), rays=512, seed=3 + rank)
    broadcast_parameters(w.graph, src=0)
    bucket = GradBucket(w.net_params)
    torch.manual_seed(100 + rank)
    w.optim.zero_grad(set_to_none=True)
    R = w.rays // w.B
    idx = torch.randperm(w.H * w.W, device=w.device)[:R]
    ret = w.graph.render(w.opt, w.data.pose, H=w.H, W=w.W, intr=w.intr, ray_idx=idx, depth_range=w.data.depth_range[0], iter=1000, mode="train")
    loss = w._photometric(ret, idx)
    loss.backward()
    local = torch.cat([p.grad.reshape(-1) for p in w.net_params]).clone()
    extra = bucket.allreduce_(average=False, extra=torch.stack([loss.detach(), torch.ones((), device=w.device)]))
    assert bucket.last_path == "flat" and bucket.collectives == 1
    summed = torch.cat([p.grad.reshape(-1) for p in w.net_params])
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)                                   # RCCL again: every rank's own gradient, summed by hand
    ref = torch.stack(gathered).sum(0)
    err = float((summed - ref).abs().max() / (ref.abs().max() + 1e-30))
    if rank == 0:
        q.put((err, float(extra[1])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs two GPUs (the 1-GPU box runs the gloo variants above)")
def test_rccl_gradient_exchange_two_gpus():
    """The exchange of bench.py --gpus N on the REAL backend (`nccl` = RCCL over xGMI), one rank per GPU: the bucket's single
    all-reduce equals the hand-summed all-gather of the ranks' own gradients.  Skipped on one-GPU boxes; any multi-GPU
    box that runs `pytest -m gpu` exercises RCCL here."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_nccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    err, count = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert err < 1e-6 and count == world


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (VERDICT r01 missing-7): it must
    spawn its own ranks and print ONE JSON line with n_gpus = 2 (here both ranks on cuda:0 over gloo)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SPARF_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rays", "512", "--config", "2",
                        "--no-roofline", "--no-cpu-baseline", "--no-other-modes", "--no-psnr"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["value"] > 0 and lines[0]["config"]["baseline_config"] == 2
    # one command, both scaling numbers: the contract line is the weak leg (512 rays per rank), followed by a
    # strong leg with the same global batch split over the ranks (256 rays per rank)
    legs = lines[0]["scaling_legs"]
    assert lines[0]["scaling"] == "weak" and abs(legs["weak"]["value"] - lines[0]["value"]) < 1e-6 * lines[0]["value"]
    assert round(legs["weak"]["rays_per_gpu_per_step"]) == 510 and round(legs["strong"]["rays_per_gpu_per_step"]) == 255      # 3 views x 170 / 85
    assert legs["strong"]["value"] > 0 and legs["strong"]["per_rank_ms_per_step"]["max"] >= legs["strong"]["per_rank_ms_per_step"]["min"]
    assert legs["strong"]["launch"].startswith("two hipGraphs"), legs["strong"]["launch"]      # the strong leg's small steps are captured


def _captured_dp_worker(rank, world, port, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "compat")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sparf_amd.parallel import GradBucket, broadcast_parameters
    from bench_workloads import Workload
    dev = torch.device("cuda:0")

    def factory(w):
        bucket = GradBucket(w.net_params + [w.graph.se3_refine])
        w.bucket = bucket
        return lambda loss: setattr(w, "last_scalars", bucket.allreduce_(extra=torch.stack([loss.detach(), torch.isnan(loss.detach()).float()])))

    w = Workload(2, "bf16x3", dev, rays=255, bucket_factory=factory, graph_capture=True, seed=3 + rank)      # different weights per rank ...
    broadcast_parameters(w.graph)                                                                          # ... made identical here
    torch.cuda.manual_seed(50 + rank)                  # each rank: its own ray shard and draws
    torch.manual_seed(50 + rank)
    step = w.capture(warmup=2)
    assert w._graph_update is not None, "a data-parallel step is captured as two graphs around the exchange"
    losses = [float(step()) for _ in range(6)]
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in w.net_params] + [w.graph.se3_refine.detach().reshape(-1)]).cpu()
    n_coll = w.bucket.collectives
    gathered = [None] * world
    dist.all_gather_object(gathered, (flat.numpy(), losses, n_coll, float(w.last_scalars[0])))
    if rank == 0:
        q.put(gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_captured_step_takes_part_in_data_parallelism():
    """the step captured as hipGraphs with a gradient bucket -- forward + backward | ONE eager all-reduce |
    clip + Adam -- on two ranks (gloo, one GPU): both ranks hold identical parameters after every replay although each renders its
    own rays (their losses differ), i.e. the exchange happens between the two graphs and on the captured gradient buffers."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_captured_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    # poll with a timeout while watching the workers: a rank that dies (capture failure, gloo init, an assert) must FAIL the test, not
    # hang the GPU suite on a queue nobody will ever write
    import queue as _queue
    import time as _time
    got, deadline = None, _time.time() + 600
    while got is None:
        try:
            got = q.get(timeout=2.0)
        except _queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or _time.time() > deadline or all(p.exitcode is not None for p in procs):
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                pytest.fail(f"data-parallel workers ended without a result (exit codes {[p.exitcode for p in procs]})")
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (p0, l0, c0, s0), (p1, l1, c1, s1) = got
    assert c0 == 1 and c1 == 1, "one all-reduce per step"
    assert np.array_equal(p0, p1), float(np.abs(p0 - p1).max())          # same reduced gradients, same deterministic updates
    assert all(np.isfinite(l0)) and all(np.isfinite(l1)) and l0 != l1    # own shards, own draws
    assert abs(s0 - (l0[-1] + l1[-1])) < 1e-5 * abs(s0)                  # the loss scalars rode along in the same message
