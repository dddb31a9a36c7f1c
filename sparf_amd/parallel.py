"""Ray-batch data parallelism (new capability; the reference is single-GPU only,
/root/reference/train_settings/default_config.py:25, README.md:48).

One process per GPU, full replicas of both networks, each rank renders its own shard of
the ray batch; the only exchange is ONE all-reduce (sum) per step over a single flat fp32
gradient bucket (2 x 530 052 floats = 4.24 MB + whatever else is registered, e.g. pose
parameters).  On a fully connected xGMI node that message is latency-bound, so there is
no bucketing/overlap machinery: RCCL's default algorithm, issued once after backward.
Backend: `nccl` (= RCCL on ROCm) on GPUs, `gloo` in the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_slice(n, rank, world):
    """Contiguous shard [lo, hi) of n items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradBucket:
    """Flat gradient bucket over a fixed parameter list."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None

    def allreduce_(self, group=None, average=True, extra=None):
        """Sum gradients over ranks (missing grads count as zero) and write them back.
        `extra`: optional 1-D tensor of scalars reduced in the same message (loss sums,
        valid counts, NaN flag); the reduced values are returned."""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        dev = self.params[0].device
        n_extra = 0 if extra is None else extra.numel()
        if self.flat is None or self.flat.numel() != self.numel + n_extra or self.flat.device != dev:
            self.flat = torch.zeros(self.numel + n_extra, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        if n_extra:
            self.flat[off:].copy_(extra.reshape(-1).to(torch.float32))
        if world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        scale = 1.0 / world if average else 1.0
        off = 0
        for p in self.params:
            n = p.numel()
            g = self.flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = (g * scale).clone()
            else:
                p.grad.copy_(g).mul_(scale)
            off += n
        return self.flat[off:].clone() if n_extra else None


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s parameters (one flat broadcast)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
