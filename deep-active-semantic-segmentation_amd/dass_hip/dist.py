"""One-process-per-GPU helpers over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The reference is single-process nn.DataParallel: every forward it broadcasts all parameters (237 MB for R101) and
gathers logits on device 0.  Here each rank owns a full replica; the only training collective is the
gradient average below, and the scoring collectives live in active_selection/base.py.
"""
import torch


class ModuleWrapper(torch.nn.Module):
    """gives a bare model the `.module` attribute the selectors and `active_train.py:440-441` expect from
    nn.DataParallel, without any replication"""

    def __init__(self, module):
        super(ModuleWrapper, self).__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def average_gradients(params, bucket_bytes=64 << 20):
    """DDP-style gradient averaging after backward: gradients are packed into flat f32 buckets, every bucket is one
    asynchronous all-reduce (they pipeline on the RCCL stream), and the averaged values are scattered back in place.
    Returns the number of buckets.  With per-GPU batches of equal size this reproduces the single-process gradient
    of the reference's DataParallel step (loss averaged over the global batch)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    bucket, size, works = [], 0, []
    cap = bucket_bytes // 4

    def flush():
        nonlocal bucket, size
        if bucket:
            flat = torch.cat([p.grad.reshape(-1) for p in bucket])
            works.append((dist.all_reduce(flat, async_op=True), flat, bucket))
            bucket, size = [], 0

    for p in params:
        if p.grad is None:
            continue
        bucket.append(p)
        size += p.grad.numel()
        if size >= cap:
            flush()
    flush()
    for work, flat, ps in works:
        work.wait()
        flat.div_(world)
        off = 0
        for p in ps:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
    return len(works)
