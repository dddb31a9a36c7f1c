"""Ray-batch data parallelism (new capability; the reference is single-GPU only,
/root/reference/train_settings/default_config.py:25, README.md:48).

One process per GPU, full replicas of both networks, each rank renders its own shard of
the ray batch; the only exchange is ONE all-reduce (sum) per step over a single flat fp32
message: both networks' gradients (2 x 530 052 floats = 4.24 MB; the HIP backward delivers
each network's 20 gradients as one flat buffer, so the message is assembled by one `cat`
of two large pieces), whatever else is registered (pose parameters) and the step's scalars
(loss, NaN flag) appended at the tail.  On a fully connected xGMI node that message is
latency-bound, so there is no bucketing/overlap machinery: RCCL's default algorithm,
issued once after backward.
Backend: `nccl` (= RCCL on ROCm) on GPUs, `gloo` in the CPU tests.
"""
import torch
import torch.distributed as dist


def shard_slice(n, rank, world):
    """Contiguous shard [lo, hi) of n items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _group_grads(grads):
    """Split gradient tensors into (flats, loose): one flat fp32 view per storage that two or more of them
    tile without gaps (or that a single large one fills), and the remaining small tensors."""
    by_storage = {}
    for g in grads:
        by_storage.setdefault(g.untyped_storage().data_ptr(), []).append(g)
    flats, loose = [], []
    for gs in by_storage.values():
        gs.sort(key=lambda g: g.storage_offset())
        ok = all(g.dtype == torch.float32 and g.is_contiguous() for g in gs)
        pos = gs[0].storage_offset()
        for g in gs:
            ok = ok and g.storage_offset() == pos
            pos += g.numel()
        if ok and (len(gs) > 1 or gs[0].numel() >= 4096):
            lo = gs[0].storage_offset()
            flats.append(torch.empty(0, dtype=torch.float32, device=gs[0].device).set_(gs[0].untyped_storage(), lo, (pos - lo,)))
        else:
            loose += gs
    return flats, loose


class GradBucket:
    """Gradient exchange over a fixed parameter list in ONE all-reduce per call.

    The message = [flat gradient buffers of the networks | remaining small gradients | extra scalars].  When the
    gradients are views into flat buffers (the HIP backward's) the message is assembled by one `cat` of a few large
    pieces and copied back piecewise (`last_path == "flat"`; a single flat buffer without extras is reduced in
    place); gradients produced tensor by tensor (any autograd graph, the CPU tests) go through the same staging
    buffer parameter by parameter (`"bucket"`)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None
        self.last_path = None
        self.collectives = 0           # all-reduces issued by the last call (always 1 when world > 1)

    def allreduce_(self, group=None, average=True, extra=None):
        """Sum (or average) gradients over ranks (missing grads count as zero) and write them
        back.  `extra`: optional 1-D tensor of scalars reduced in the same exchange (loss sums,
        valid counts, NaN flag); the reduced values are returned (summed, never averaged)."""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        dev = self.params[0].device
        scale = 1.0 / world if average else 1.0
        n_extra = 0 if extra is None else extra.numel()
        grads = [p.grad for p in self.params]
        self.collectives = 0

        def reduce_(t):
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                self.collectives += 1

        if all(g is not None for g in grads):
            flats, loose = _group_grads(grads)
            if flats and len(flats) + len(loose) <= 8:
                self.last_path = "flat"
                if len(flats) == 1 and not loose and not n_extra:
                    reduce_(flats[0])
                    if scale != 1.0:
                        flats[0].mul_(scale)
                    return None
                pieces = flats + [g.reshape(-1).to(torch.float32) for g in loose]
                tail = [extra.reshape(-1).to(torch.float32)] if n_extra else []
                msg = torch.cat(pieces + tail)                       # one launch: a few large pieces
                reduce_(msg)
                off = 0
                for dst in flats + loose:
                    n = dst.numel()
                    if scale != 1.0:
                        torch.mul(msg[off:off + n].view_as(dst), scale, out=dst)
                    else:
                        dst.copy_(msg[off:off + n].view_as(dst))
                    off += n
                return msg[off:].clone() if n_extra else None
        self.last_path = "bucket"
        if self.flat is None or self.flat.numel() != self.numel + n_extra or self.flat.device != dev:
            self.flat = torch.zeros(self.numel + n_extra, dtype=torch.float32, device=dev)
        pieces = [(g.reshape(-1).to(torch.float32) if g is not None else torch.zeros(p.numel(), dtype=torch.float32, device=dev))
                  for p, g in zip(self.params, grads)]
        if n_extra:
            pieces.append(extra.reshape(-1).to(torch.float32))
        torch.cat(pieces, out=self.flat)
        reduce_(self.flat)
        off = 0
        for p in self.params:
            n = p.numel()
            g = self.flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = (g * scale).clone()
            else:
                p.grad.copy_(g)
                if scale != 1.0:
                    p.grad.mul_(scale)
            off += n
        return self.flat[off:].clone() if n_extra else None


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s parameters (one flat broadcast)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    tensors = list(module.parameters()) + list(module.buffers())
    with torch.no_grad():
        flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))       # on the parameter itself: bumps its version counter
            off += n
    for m in module.modules():                          # and tell version-keyed caches explicitly
        if hasattr(m, "weights_changed"):
            m.weights_changed()
