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


class GradientAverager(object):
    """Gradient averaging overlapped with backward (what DistributedDataParallel's reducer does, on RCCL):
    parameters are assigned to flat f32 buckets in REVERSE registration order (roughly the order backward produces
    their gradients); a post-accumulate-grad hook counts a bucket's gradients in, and a full bucket is packed and
    all-reduced asynchronously while backward keeps running on the compute stream.  Buckets are launched strictly in
    index order, so every rank issues the same collectives in the same sequence whatever the order its hooks fire in.
    `finish()` (after backward) launches what is left, waits, scales by 1/world and scatters the averages back.

        averager = GradientAverager(params)          # once
        loss.backward(); averager.finish(); optimizer.step()

    With one process (or no process group) it does nothing."""

    def __init__(self, params, bucket_bytes=64 << 20):
        import torch.distributed as dist

        self.dist = dist
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.params = [p for p in params if p.requires_grad]
        self.buckets, self.where = [], {}
        if not self.active:
            return
        self.world = dist.get_world_size()
        cap, cur, size = bucket_bytes // 4, [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel()
            if size >= cap:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        for bi, ps in enumerate(self.buckets):
            for p in ps:
                self.where[id(p)] = bi
                p.register_post_accumulate_grad_hook(self._hook)
        self._reset()

    def _reset(self):
        self.pending = [len(ps) for ps in self.buckets]
        self.next = 0
        self.works = []

    def _hook(self, p):
        bi = self.where[id(p)]
        self.pending[bi] -= 1
        self._launch_ready()

    def _launch_ready(self):
        while self.next < len(self.buckets) and self.pending[self.next] <= 0:
            self._launch(self.next)
            self.next += 1

    def _launch(self, bi):
        ps = [p for p in self.buckets[bi] if p.grad is not None]
        if not ps:
            self.works.append(None)
            return
        flat = torch.cat([p.grad.reshape(-1).float() for p in ps])
        self.works.append((self.dist.all_reduce(flat, async_op=True), flat, ps))

    def finish(self):
        """call after backward on every rank; returns the number of all-reduces issued"""
        if not self.active:
            return 0
        # parameters that received no gradient this step never fired: their buckets go out now, in index order
        while self.next < len(self.buckets):
            self._launch(self.next)
            self.next += 1
        n = 0
        for item in self.works:
            if item is None:
                continue
            work, flat, ps = item
            work.wait()
            flat.div_(self.world)
            off = 0
            for p in ps:
                k = p.grad.numel()
                p.grad.copy_(flat[off:off + k].view_as(p.grad))
                off += k
            n += 1
        self._reset()
        return n
