"""ActiveSelectionBase -- mirror of active_selection/base.py:1-6 plus the two pieces every selector
shares in this build: how the pool is read and how it is sharded over ranks.

Data layer.  The reference selectors build DataLoader(PathsDataset(self.env, keys, crop, labels),
batch_size, shuffle=False, num_workers=0) themselves (mc_dropout.py:180-181).  Here the loader comes
from `loader_factory(keys, include_labels)` when one is injected (tests, bench: synthetic pools
resident in HBM), else from `dataloaders.dataset.paths_dataset.pool_loader` -- this build's
PathsDataset (records fetched by a thread pool, PIL-exact resize / crop / normalise on the GPU).

Multi-GPU.  Pool scoring is image-independent (eval-mode BN), so with torch.distributed initialised
(one process per GPU, RCCL) each rank scores a contiguous shard of the key list and the per-image
scores are all-gathered (padded to equal length); every rank then runs the same stable sort and
returns the same selection.  Region scoring shards the same way: each rank builds the box-sum score
maps of its shard, the global min / max are two one-float all-reduces, the normalised maps are
all-gathered (2975 x 385^2 f32 = 1.76 GB over xGMI, tens of milliseconds next to a minute of forward
passes) and the greedy square NMS -- a global sequential argmax by definition -- runs on every rank
over the full set, so all ranks return identical regions.
"""
import torch


def _dist():
    import torch.distributed as dist

    return dist if dist.is_available() and dist.is_initialized() else None


def shard_bounds(n_items, rank, world_size):
    """contiguous, balanced split: the first (n % world) ranks hold one extra item"""
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_rows(local, n_total, device=None):
    """local: [n_local, ...] tensor of this rank's shard (shard_bounds order) -> [n_total, ...] on every rank.
    One all_gather of equal-size padded buffers (RCCL over xGMI when the backend is nccl)."""
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (n_total + world - 1) // world
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    parts = []
    for r in range(world):
        s, e = shard_bounds(n_total, r, world)
        parts.append(out[r][: e - s])
    return torch.cat(parts, dim=0)


def merged_batches(loader, merge):
    """`merge` consecutive batches of a pool loader as one: dicts are concatenated key by key, bare tensors along dim 0.  Pool scoring
    is image-independent (eval-mode BN), so how many loader batches share a scoring forward changes no score -- but the encoder's
    33 x 33 layers fill the chip only from ~16 images on (R101 513^2: MC-dropout 549 -> 620 pool images/s, DASS_SCORE_MERGE)."""
    if merge <= 1:
        for sample in loader:
            yield sample
        return
    held = []
    for sample in loader:
        if held and not _mergeable(held[0], sample):
            for h in held:     # (crop_size = -1 pools: batches of another spatial size go through on their own)
                yield h
            held = []
        held.append(sample)
        if len(held) == merge:
            yield _cat_samples(held)
            held = []
    if len(held) > 1 and all(_mergeable(held[0], h) for h in held[1:]):
        yield _cat_samples(held)
    else:
        for h in held:
            yield h


def _mergeable(a, b):
    """may two loader batches share a forward?  same container type, same keys, tensors of equal trailing shape and dtype"""
    if isinstance(a, dict) != isinstance(b, dict):
        return False
    if isinstance(a, dict):
        if a.keys() != b.keys():
            return False
        pairs = [(a[k], b[k]) for k in a]
    else:
        pairs = [(a, b)]
    for u, v in pairs:
        if torch.is_tensor(u) != torch.is_tensor(v):
            return False
        if torch.is_tensor(u) and (u.shape[1:] != v.shape[1:] or u.dtype != v.dtype or u.device != v.device):
            return False
    return True


def _cat_samples(samples):
    """tensors are concatenated along dim 0; any other value of a dict sample (names, ids: lists / tuples / scalars) is carried as the
    concatenated list, in batch order"""
    if len(samples) == 1:
        return samples[0]
    if isinstance(samples[0], dict):
        out = {}
        for k in samples[0]:
            vals = [s[k] for s in samples]
            if torch.is_tensor(vals[0]):
                out[k] = torch.cat(vals, dim=0)
            else:
                flat = []
                for v in vals:
                    flat.extend(list(v) if isinstance(v, (list, tuple)) else [v])
                out[k] = flat
        return out
    return torch.cat(list(samples), dim=0)


def score_merge(batch_size=None, most=2):
    """loader batches per scoring forward.  DASS_SCORE_MERGE when set (1 = the loader's own batches); otherwise the largest factor <= `most` that
    keeps the merged forward at or below 8 x most images, 1 above: a user who sized `dataloader_batch_size` to the memory of a 769^2 crop or a large
    T is not handed a multiple of it behind their back.  Measured on R101 with the driver's batch of 8 (tools/score_merge_sweep.sh), 1 / 2 / 3 / 4
    batches per forward: core-set features 1274 / 1448 / 1564-1581 / 1441 pool images/s -> the feature pass asks for most = 3; MC-dropout 565 / 631 /
    637-644 / 626 at 513^2 but 298 -> 285 at 769^2 with three -> the T-pass selectors stay at most = 2 (the encoder's 33 x 33 layers fill the chip
    from ~16 images on; beyond 24 the tile counts quantise worse again)."""
    import os

    env = os.environ.get("DASS_SCORE_MERGE")
    if env is not None:
        return max(1, int(env))
    if batch_size is None:
        return 2
    for m in range(int(most), 1, -1):
        if m * int(batch_size) <= 8 * int(most):
            return m
    return 1


class ActiveSelectionBase:

    def __init__(self, dataset_lmdb_env, crop_size, dataloader_batch_size, loader_factory=None, shard=True):
        self.crop_size = crop_size
        self.dataloader_batch_size = dataloader_batch_size
        self.env = dataset_lmdb_env
        self.loader_factory = loader_factory
        self.shard = shard

    # ---- pool access
    def make_loader(self, images, include_labels):
        if self.loader_factory is not None:
            return self.loader_factory(images, include_labels)
        from dataloaders.dataset import paths_dataset  # this build's device-side PathsDataset (or the caller's own package)

        if hasattr(paths_dataset, "pool_loader"):  # records prefetched by threads, resize / crop / normalise on the GPU
            return paths_dataset.pool_loader(self.env, images, self.crop_size, include_labels, self.dataloader_batch_size)
        from torch.utils.data import DataLoader

        return DataLoader(paths_dataset.PathsDataset(self.env, images, self.crop_size, include_labels=include_labels),
                          batch_size=self.dataloader_batch_size, shuffle=False, num_workers=0)

    # ---- sharding
    def local_slice(self, images):
        dist = _dist()
        if not self.shard or dist is None or dist.get_world_size() == 1:
            return list(images), 0
        s, e = shard_bounds(len(images), dist.get_rank(), dist.get_world_size())
        return list(images[s:e]), s

    def gather(self, local_rows, n_total):
        if not self.shard:
            return local_rows
        return all_gather_rows(local_rows, n_total)

    def global_minmax(self, mm):
        """mm: device tensor [min, max] of this rank's shard -> the extrema over every rank's shard (two all-reduces of
        one float: mc_dropout.py:152-155 normalises the region score maps of the WHOLE pool with one min and one max)"""
        dist = _dist()
        if not self.shard or dist is None or dist.get_world_size() == 1:
            return mm
        lo, hi = mm[0:1].clone(), mm[1:2].clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return torch.cat((lo, hi))

    @staticmethod
    def unwrap(model):
        """selectors receive the DataParallel/DDP-wrapped model (active_train.py:82-85, core_set.py:44)"""
        return model.module if hasattr(model, "module") else model

    @staticmethod
    def label_mask(label, num_classes):
        return (label < 0) | (label >= num_classes)
