"""One-process-per-GPU helpers over torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The reference is single-process nn.DataParallel: every forward it broadcasts all parameters (237 MB for R101) and
gathers logits on device 0.  Here each rank owns a full replica; the only training collective is the
gradient average below, and the scoring collectives live in active_selection/base.py.
"""
import os
import torch


class ModuleWrapper(torch.nn.Module):
    """gives a bare model the `.module` attribute the selectors and `active_train.py:440-441` expect from
    nn.DataParallel, without any replication"""

    def __init__(self, module):
        super(ModuleWrapper, self).__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def world_size():
    import torch.distributed as dist

    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def collectives_on():
    """do the training collectives run?  A process group with more than one rank -- or, DASS_DIST_FORCE=1, a group of ONE rank: the rehearsal of
    the multi-process step on a one-GPU box with the real backend (RCCL communicator, its watchdog thread, hipGraph capture beside it);
    every collective then runs over a single rank and changes no value"""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("DASS_DIST_FORCE") == "1"


class _SumOverRanks(torch.autograd.Function):
    """y = sum over ranks of x (one all-reduce).  The true derivative dy/dx_r is 1; this backward returns
    world * g because the gradients of every rank are AVERAGED afterwards (GradientAverager / average_gradients /
    torch DDP all divide the all-reduced sum by world): (1/world) * sum_r J_r^T (world * g) = sum_r J_r^T g, the
    single-process gradient."""

    @staticmethod
    def forward(ctx, x):
        import torch.distributed as dist

        y = x.detach().clone()
        dist.all_reduce(y)
        ctx.world = dist.get_world_size()
        return y

    @staticmethod
    def backward(ctx, g):
        return g * ctx.world


def sum_over_ranks(x, differentiable=True):
    """x summed over all ranks (identity without a process group).  differentiable=True: see _SumOverRanks."""
    if not collectives_on():
        return x
    if differentiable and x.requires_grad:
        return _SumOverRanks.apply(x)
    import torch.distributed as dist

    y = x.detach().clone()
    dist.all_reduce(y)
    return y


def global_batch_mean(local_sum, local_count, n_local):
    """The reference computes its loss on device 0 over the GATHERED global batch (nn.DataParallel,
    active_train.py:104-105): nn.CrossEntropyLoss(reduction='mean') over every valid pixel of every replica's
    images, then `/ n` with n the GLOBAL batch size (utils/loss.py:39-51).  One process per GPU reproduces it from
    each rank's numerator (sum of w*nll over its valid pixels), denominator (sum of w over them) and image count:
    one all-reduce of three floats.  -> (mean over the global batch [same value on every rank; autograd],
    global batch size).  Ranks may hold different numbers of valid pixels and of images."""
    if not collectives_on():
        return local_sum / local_count, n_local
    packed = torch.stack((local_count.detach().float().reshape(()),
                          torch.as_tensor(float(n_local), dtype=torch.float32, device=local_sum.device)))
    tot = sum_over_ranks(packed, differentiable=False)
    total_sum = sum_over_ranks(local_sum)
    return total_sum / tot[0], int(round(float(tot[1])))


class StaticGlobalBatch(object):
    """The denominators of the global-batch loss, exchanged AHEAD of the step: they depend on the labels only (valid-pixel weight sum,
    image count), so a step whose forward / backward must not hold a collective -- a hipGraph replay (dass_hip/graph.py) -- can
    normalise with device scalars that `exchange(target)` refreshed eagerly before it.  utils.loss.SegmentationLosses(global_batch=True)
    uses them after `criterion.use_static_global(device)`: each rank's loss is then  world * (its numerator) / (global count) / (global
    batch)  -- the same gradient scaling as global_batch_mean's backward -- and the mean of the ranks' values is the global loss."""

    def __init__(self, device, ignore_index=255, weight=None):
        self.ignore_index, self.weight = ignore_index, weight
        self.inv_count = torch.ones((), dtype=torch.float32, device=device)   # world / sum over ranks of the valid-pixel weights
        self.n_global = torch.ones((), dtype=torch.float32, device=device)    # global batch size

    @torch.no_grad()
    def exchange(self, target, n_local=None):
        import torch.distributed as dist

        valid = target != self.ignore_index
        if self.weight is None:
            cnt = valid.sum().float()
        else:
            w = self.weight.to(device=target.device, dtype=torch.float32)
            cnt = (w[target.clamp(0, w.numel() - 1).long()] * valid).sum()
        packed = torch.stack((cnt, torch.as_tensor(float(target.shape[0] if n_local is None else n_local), dtype=torch.float32, device=target.device)))
        world = world_size()
        if collectives_on():
            dist.all_reduce(packed)
        self.inv_count.copy_(float(world) / packed[0])
        self.n_global.copy_(packed[1])


def _dense_extent(g):
    """(first element, element count) of g inside its storage when g occupies ONE dense run of memory in some dimension order (an OIHW
    view of KRSC memory does), else None"""
    if g.numel() == 0:
        return None
    span = 1 + sum((n - 1) * st for n, st in zip(g.shape, g.stride()) if n > 1)
    if span != g.numel() or any(st < 0 for st in g.stride()):
        return None
    return g.storage_offset(), g.numel()


def average_gradients(params, bucket_bytes=64 << 20):
    """DDP-style gradient averaging after backward.  Returns the number of all-reduces issued.
      * Gradients that ALREADY share a storage -- the conv weight gradients are slices of one zeroed arena (ops._zeroed_dw), 99.7 % of
        R101's 237 MB -- are all-reduced IN PLACE over the covering range of that storage: no pack, no scatter, one collective per arena
        (the gaps between slices are alignment padding that stays zero).
      * The rest (BN affine parameters, biases: a few hundred small tensors) ride in flat f32 buckets built and scattered by multi-tensor
        ops (one cat, one _foreach_copy_ per bucket) -- per-parameter copies cost the host ~5 us each, 1.6 ms per step for R101.
      * A parameter without a gradient on this rank sends the step down the general path below (zeros + per-parameter "fired" flags, so
        bucket sizes never depend on which parameters fired).
    The average over ranks equals the reference's single-process DataParallel gradient when the loss is the global-batch loss of
    utils.loss.SegmentationLosses (global_batch_mean above / StaticGlobalBatch: numerators, valid-pixel counts and batch sizes are
    exchanged, and the backward pre-multiplies by world)."""
    import torch.distributed as dist

    if not collectives_on():
        return 0
    world = dist.get_world_size()
    ps = [p for p in params if p.requires_grad]
    if any(p.grad is None for p in ps):
        return _average_gradients_bucketed(ps, bucket_bytes)
    # in place only over RCCL / NCCL: gloo stages a CUDA tensor through the host, and its path for a VIEW into a large storage turned
    # out pathological (0.7 s for the 237 MB range as one call, 6.8 s as four, against 60 ms for four packed 64 MB buckets)
    inplace_ok = dist.get_backend() == "nccl" or os.environ.get("DASS_DDP_INPLACE") == "1"
    by_storage, rest = {}, []
    for p in ps:
        g = p.grad
        ext = _dense_extent(g) if g.dtype == torch.float32 else None
        if ext is None or not inplace_ok:
            rest.append((g, ext))
        else:
            by_storage.setdefault(g.untyped_storage().data_ptr(), []).append((g, ext))
    works = []
    for items in by_storage.values():
        lo = min(e[0] for _, e in items)
        hi = max(e[0] + e[1] for _, e in items)
        used = sum(e[1] for _, e in items)
        if len(items) < 2 or (hi - lo) * 4 < (1 << 20) or used < 0.9 * (hi - lo):
            rest.extend(items)   # a lone tensor, a small group, or slices too far apart: through the buckets
            continue
        g0 = items[0][0]
        flat = torch.empty((0,), dtype=torch.float32, device=g0.device).set_(g0.untyped_storage(), lo, (hi - lo,))
        works.append((dist.all_reduce(flat, async_op=True), flat, None, None))
    cap, cur, size = bucket_bytes // 4, [], 0
    buckets = []
    for g, ext in rest:
        # a dense f32 gradient travels as the flat view of its own memory (an OIHW view of KRSC memory included: no re-layout copy)
        v = (torch.empty((0,), dtype=torch.float32, device=g.device).set_(g.untyped_storage(), ext[0], (ext[1],)) if ext is not None else None)
        cur.append((g, v))
        size += g.numel()
        if size >= cap:
            buckets.append(cur)
            cur, size = [], 0
    if cur:
        buckets.append(cur)
    for gv in buckets:
        srcs = [v if v is not None else g.contiguous().reshape(-1).float() for g, v in gv]
        flat = torch.cat(srcs) if len(srcs) > 1 else srcs[0].clone()
        works.append((dist.all_reduce(flat, async_op=True), flat, gv, [t.numel() for t in srcs]))
    for work, flat, gv, sizes in works:
        work.wait()
        flat.div_(world)
        if gv is not None:
            parts = flat.split(sizes)
            direct = [(v, t) for (g, v), t in zip(gv, parts) if v is not None]
            if direct:
                torch._foreach_copy_([v for v, _ in direct], [t for _, t in direct])
            for (g, v), t in zip(gv, parts):
                if v is None:
                    g.copy_(t.view_as(g))
    return len(works)


def _average_gradients_bucketed(ps, bucket_bytes=64 << 20):
    """the general form: every parameter packed into flat buckets, missing gradients as zeros + "fired" flags (see _pack / _unpack)"""
    import torch.distributed as dist

    world = dist.get_world_size()
    bucket, size, works = [], 0, []
    cap = bucket_bytes // 4

    def flush():
        nonlocal bucket, size
        if bucket:
            flat = _pack(bucket)
            works.append((dist.all_reduce(flat, async_op=True), flat, bucket))
            bucket, size = [], 0

    for p in ps:
        bucket.append(p)
        size += p.numel()
        if size >= cap:
            flush()
    flush()
    for work, flat, bps in works:
        work.wait()
        flat.div_(world)
        _unpack(flat, bps)
    return len(works)


def _pack(ps):
    """flat f32 bucket of the gradients of ps, in order, followed by one "fired" flag per parameter.  A parameter that
    received no gradient on this rank rides as zeros with flag 0, so every rank packs the same number of elements
    whichever parameters fired locally (data-dependent branches, set_to_none=True)."""
    dev = ps[0].device
    parts = [(p.grad.reshape(-1).float() if p.grad is not None else torch.zeros((p.numel(),), dtype=torch.float32, device=dev))
             for p in ps]
    parts.append(torch.tensor([0.0 if p.grad is None else 1.0 for p in ps], dtype=torch.float32, device=dev))
    return torch.cat(parts)


def _unpack(flat, ps):
    """flat: the all-reduced bucket already divided by world.  A parameter that fired on NO rank keeps grad None (the
    optimizer skips it, as in the single-process step); one that fired elsewhere only receives the average."""
    # the flags are read back (a device sync) only when some local gradient is missing: the common step never waits
    fired = flat[flat.numel() - len(ps):].tolist() if any(p.grad is None for p in ps) else [1.0] * len(ps)
    off = 0
    for p, f in zip(ps, fired):
        k = p.numel()
        if p.grad is not None:
            p.grad.copy_(flat[off:off + k].view_as(p.grad))
        elif f > 0.0:
            p.grad = flat[off:off + k].view_as(p).clone()
        off += k


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
        self.active = collectives_on()
        self.params = [p for p in params if p.requires_grad]
        self.buckets, self.where = [], {}
        if not self.active:
            return
        self.world = dist.get_world_size()
        # (deferred weight gradients are flushed in chunks during backward -- ops.set_wgrad_chunk, default 16 layers -- so the first
        # buckets fill, and their all-reduce starts, long before the pass ends)
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
        self.seen = set()

    def _hook(self, p):
        from . import ops

        if ops.wgrad_pending(p):  # its gradient is still queued for the grouped launch (ops._wgrad_flush calls again when it is set)
            return
        if id(p) in self.seen:    # counted once per pass: a weight used through the deferred AND the generic path (or a module
            return                # applied twice) fires this hook more than once
        self.seen.add(id(p))
        bi = self.where[id(p)]
        self.pending[bi] -= 1
        self._launch_ready()

    def _launch_ready(self):
        while self.next < len(self.buckets) and self.pending[self.next] <= 0:
            self._launch(self.next)
            self.next += 1

    def _launch(self, bi):
        ps = self.buckets[bi]
        flat = _pack(ps)  # rank-invariant size: missing gradients ride as zeros
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
        for work, flat, ps in self.works:
            work.wait()
            flat.div_(self.world)
            _unpack(flat, ps)
            n += 1
        self._reset()
        return n
