"""Ray-batch data parallelism on CPU (gloo, world size 2): the sharded step must equal the
single-process step on the same global ray set.  The compute inside each rank is the
oracle (test infrastructure) -- what is under test is sparf_amd.parallel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sparf_amd.parallel import GradBucket, broadcast_parameters, shard_slice


def test_shard_slice_partitions():
    for n in (0, 1, 7, 4096, 4095):
        for world in (1, 2, 3, 8):
            spans = [shard_slice(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_problem():
    from oracle import nerf_oracle as O
    from tests.golden.recipe import small_opt, make_state_dict
    opt = small_opt(nerf=dict(fine_sampling=False))
    rs = np.random.RandomState(0)
    R, N = 12, 8
    center = torch.from_numpy(rs.uniform(-0.5, 0.5, size=(1, R, 3)).astype(np.float32)) + torch.tensor([0.0, 0.0, -3.0])
    ray = torch.from_numpy(rs.uniform(-0.3, 0.3, size=(1, R, 3)).astype(np.float32)) + torch.tensor([0.0, 0.0, 1.0])
    jitter = torch.from_numpy(rs.uniform(size=(1, R, N, 1)).astype(np.float32))
    target = torch.from_numpy(rs.uniform(size=(1, R, 3)).astype(np.float32))
    return O, opt, make_state_dict, center, ray, jitter, target


class TinyNet(torch.nn.Module):
    def __init__(self, sd):
        super().__init__()
        self.p = torch.nn.ParameterDict({k.replace(".", "_"): torch.nn.Parameter(v.clone()) for k, v in sd.items() if k != "progress"})
        self.register_buffer("progress", sd["progress"].clone())

    def sd(self):
        d = {k.replace("_weight", ".weight").replace("_bias", ".bias").replace("mlp_feat_", "mlp_feat.").replace("mlp_rgb_", "mlp_rgb."): v
             for k, v in self.p.items()}
        d["progress"] = self.progress
        return d


def _loss_sum(O, opt, net, center, ray, jitter, target, lo, hi):
    out = O.render(opt, net.sd(), None, center[:, lo:hi], ray[:, lo:hi], [1.2, 5.2], mode="train", jitter=jitter[:, lo:hi])
    return ((out["rgb"] - target[:, lo:hi]) ** 2).sum()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    O, opt, make_sd, center, ray, jitter, target = _make_problem()
    net = TinyNet(make_sd(opt, 100 + rank))          # different init per rank ...
    broadcast_parameters(net, src=0)                 # ... made identical here
    R = ray.shape[1]
    lo, hi = shard_slice(R, rank, world)
    loss = _loss_sum(O, opt, net, center, ray, jitter, target, lo, hi) / (R * 3)   # global normaliser
    loss.backward()
    bucket = GradBucket(list(net.parameters()))
    extra = bucket.allreduce_(average=False, extra=torch.tensor([loss.item(), float(hi - lo)]))
    assert bucket.last_path == "bucket" and bucket.collectives == 1      # autograd made separate gradient tensors; still ONE all-reduce
    # second exchange, the way the HIP backward hands gradients over: views into ONE flat
    # buffer per network -> reduced in place, no staging copies
    plist = list(net.parameters())
    flat = torch.cat([torch.full((p.numel(),), float(rank + 1 + i)) for i, p in enumerate(plist)])
    off = 0
    for p in plist:
        p2 = flat[off:off + p.numel()].view_as(p)
        off += p.numel()
        p.grad2 = p2
    saved = [p.grad for p in plist]
    for p in plist:
        p.grad = p.grad2
    b2 = GradBucket(plist)
    b2.allreduce_(average=True)
    assert b2.last_path == "flat" and b2.collectives == 1
    for i, p in enumerate(plist):                    # mean over ranks of (rank + 1 + i)
        assert torch.allclose(p.grad, torch.full_like(p.grad, (world + 1) / 2 + i)), i
    for p, g in zip(plist, saved):
        p.grad = g
    if rank == 0:
        # numpy payloads: torch tensors travel by fd-passing and need the producer alive
        q.put(({k: v.grad.numpy().copy() for k, v in net.p.items()}, extra.numpy().copy(),
               {k: v.detach().numpy().copy() for k, v in net.p.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_step_equals_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    grads, extra, params0 = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    O, opt, make_sd, center, ray, jitter, target = _make_problem()
    net = TinyNet(make_sd(opt, 100))
    for k, v in net.p.items():                       # broadcast gave every rank rank-0's weights
        assert np.array_equal(v.detach().numpy(), params0[k])
    R = ray.shape[1]
    loss = _loss_sum(O, opt, net, center, ray, jitter, target, 0, R) / (R * 3)
    loss.backward()
    assert abs(float(extra[0]) - loss.item()) < 1e-6 and float(extra[1]) == R
    for k, v in net.p.items():
        scale = v.grad.abs().max().item() + 1e-12
        assert np.abs(grads[k] - v.grad.numpy()).max() < 2e-5 * scale, k


def test_bucket_single_process_is_identity():
    lin = torch.nn.Linear(3, 2)
    lin(torch.ones(1, 3)).sum().backward()
    g = [p.grad.clone() for p in lin.parameters()]
    GradBucket(list(lin.parameters())).allreduce_()
    assert all(torch.equal(a, p.grad) for a, p in zip(g, lin.parameters()))


def _leg_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    # rank 0: 20 steps of 4096 rays in 0.10 s; rank 1: the same work in 0.16 s (the straggler sets the job's time)
    leg = bench.reduce_leg(0.10 if rank == 0 else 0.16, 20 * 4096, 20, world, torch.device("cpu"))
    strong = bench.reduce_leg(0.03 + 0.01 * rank, 30 * 2048, 30, world, torch.device("cpu"))
    if rank == 0:
        q.put((leg, strong))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_legs_reduce_over_ranks():
    """bench.py --gpus N prints a weak and a strong leg from one command; each leg's whole-job value is
    the rays ALL ranks rendered over the SLOWEST rank's time, as the contract defines `value` (world size 2, gloo)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_leg_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    weak, strong = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert abs(weak["value"] - 2 * 20 * 4096 / 0.16) < 1e-6 * weak["value"] and abs(weak["ms_per_step"] - 8.0) < 1e-9
    assert weak["per_rank_ms_per_step"] == dict(min=5.0, max=8.0) and weak["rays_per_step_all_ranks"] == 8192
    assert abs(strong["value"] - 2 * 30 * 2048 / 0.04) < 1e-6 * strong["value"]
    import bench
    one = bench.reduce_leg(0.5, 1000, 10, 1, torch.device("cpu"))
    assert one["value"] == 2000.0 and one["per_rank_ms_per_step"] is None
