"""PathsDataset on the device -- mirror of dataloaders/dataset/paths_dataset.py:8-52.

Same constructor (`env, paths, crop_size, include_labels=False`), `__len__` and `__getitem__` contract: item = the LMDB value
of key `paths[i]` (`pickle(np.uint8[H, W, 4])`, RGB + label) resized / cropped / normalised to
`{'image': f32 [3, S, S], 'label': f32 [S, S]}` (include_labels) or the bare image tensor.  Differences: the pixels are
processed by libdass_hip (csrc/pool_reader.hip) and the tensors live on the GPU, where the selectors want them; the values
are those of the reference's host pipeline (tests/golden/pool_reader.npz), bit for bit up to the final f32 normalisation.

`env` is whatever the caller opened: an `lmdb.Environment` in a real run (the module is the caller's dependency, exactly as
in the reference; it is not needed to import this file) or any object with the same read protocol
(`with env.begin(write=False) as txn: txn.get(key)`), e.g. `DictEnv` below for in-memory pools.

`pool_loader(...)` replaces `DataLoader(PathsDataset(...), batch_size, shuffle=False, num_workers=0)` of
mc_dropout.py:180-181: records are fetched + unpickled by a small thread pool a few batches ahead of the GPU (the reference
decodes and resizes each 8 MB frame on one host core between forward passes), batches come out in order.
"""
import contextlib
import ctypes
import pickle
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from dataloaders import custom_transforms as tr


class DictEnv(object):
    """in-memory stand-in for an lmdb.Environment (read protocol only): {key bytes: pickled record bytes}"""

    def __init__(self, records):
        self.records = records

    @contextlib.contextmanager
    def begin(self, write=False):
        assert not write
        yield self

    def get(self, key):
        return self.records.get(key)


def _device_tables(key, build, device):
    cache = _device_tables.cache.setdefault(str(device), {})
    if key not in cache:
        cache[key] = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in build())
    return cache[key]


_device_tables.cache = {}


class PathsDataset(torch.utils.data.Dataset):

    def __init__(self, env, paths, crop_size, include_labels=False, device=None):
        self.env = env
        self.paths = paths
        self.crop_size = crop_size
        self.include_labels = include_labels
        self.base_size = 512
        self.device = device

    def __len__(self):
        return len(self.paths)

    def read_record(self, index):
        """the host half: LMDB value -> uint8 [H, W, 4] numpy array"""
        with self.env.begin(write=False) as txn:
            loaded_npy = pickle.loads(txn.get(self.paths[index]))
        assert loaded_npy.dtype == np.uint8 and loaded_npy.ndim == 3 and loaded_npy.shape[2] == 4, "record must be uint8 [H, W, 4]"
        return np.ascontiguousarray(loaded_npy)

    def transform(self, record):
        """the device half: uint8 [H, W, 4] (numpy or device tensor) -> sample"""
        from dass_hip import ops
        from dass_hip._lib import check, lib

        if not torch.cuda.is_available():
            raise RuntimeError("PathsDataset resizes on the GPU (dass_resample_bilinear_u8); there is no CPU fallback")
        dev = torch.device(self.device) if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        rec = record if torch.is_tensor(record) else torch.from_numpy(record)
        rec = rec.to(dev, non_blocking=True).contiguous()
        h, w = int(rec.shape[0]), int(rec.shape[1])
        if self.crop_size == -1:
            oh, ow, y0, x0 = tr.scale_with_padding(h, w, self.base_size)
            size, oy0, ox0 = self.base_size, y0, x0
            divide, f64_chain = (1, 1) if self.include_labels else (0, 0)  # the image-only chain hands torchvision's ToTensor a FLOAT array: no / 255
        else:
            oh, ow, y1, x1 = tr.fix_scale_crop(h, w, self.crop_size)
            size, oy0, ox0 = self.crop_size, -y1, -x1
            divide, f64_chain = (1, 1) if self.include_labels else (1, 0)
        xt = _device_tables(("b", w, ow), lambda: tr.resample_tables(w, ow), dev)
        yt = _device_tables(("b", h, oh), lambda: tr.resample_tables(h, oh), dev)
        tmp = torch.empty((h, ow, 3), dtype=torch.uint8, device=dev)
        img = torch.empty((oh, ow, 3), dtype=torch.uint8, device=dev)
        p, st = ops._p, ops._stream()
        check(lib.dass_resample_bilinear_u8(p(rec), h, w, 4, p(tmp), p(img), oh, ow, p(xt[0]), p(xt[1]), p(xt[2]), int(xt[2].shape[1]),
                                            p(yt[0]), p(yt[1]), p(yt[2]), int(yt[2].shape[1]), st), "dass_resample_bilinear_u8")
        out_img = torch.empty((3, size, size), dtype=torch.float32, device=dev)
        out_lab = yi = xi = None
        if self.include_labels:
            out_lab = torch.empty((size, size), dtype=torch.float32, device=dev)
            (yi,) = _device_tables(("n", h, oh), lambda: (tr.nearest_table(h, oh),), dev)
            (xi,) = _device_tables(("n", w, ow), lambda: (tr.nearest_table(w, ow),), dev)
        check(lib.dass_pool_finalize(p(img), oh, ow, p(rec), w, 4, p(yi), p(xi), oy0, ox0, size, divide, f64_chain, p(out_img),
                                     p(out_lab), st), "dass_pool_finalize")
        if self.include_labels:
            return {'image': out_img, 'label': out_lab}
        return out_img

    def __getitem__(self, index):
        return self.transform(self.read_record(index))


def pool_loader(env, paths, crop_size, include_labels, batch_size, workers=4, ahead=3, device=None):
    """in-order batches of the pool: {'image': [B,3,S,S], 'label': [B,S,S]} (include_labels) or image batches, on the GPU.
    `workers` threads fetch + unpickle records up to `ahead` batches in front of the consumer and drop them into PINNED
    staging buffers, so the 8 MB host->device copy of a frame is an asynchronous DMA (from pageable memory it is a
    blocking staged copy on the consumer's thread: measured 2x the frames per second).  A staging buffer returns to the
    pool once the copy out of it has completed (event per buffer)."""
    import queue

    ds = PathsDataset(env, paths, crop_size, include_labels, device=device)
    n = len(ds)
    if not torch.cuda.is_available():
        raise RuntimeError("pool_loader feeds the GPU resize path (dass_resample_bilinear_u8); there is no CPU fallback")
    free = queue.Queue()
    n_staging = (ahead + 1) * batch_size + max(1, workers)
    staging = {}  # record shape -> buffers made so far (pools are homogeneous: one shape in practice)

    def stage(index):
        rec = ds.read_record(index)
        try:
            slot = free.get_nowait()
        except queue.Empty:
            made = staging.setdefault(rec.shape, [0])
            if made[0] < n_staging:
                made[0] += 1
                slot = (torch.empty(rec.shape, dtype=torch.uint8, pin_memory=True), torch.cuda.Event())
            else:
                # cannot happen while at most (ahead + 1) batches are staged or in flight; never block a fetch thread on the
                # consumer (an abandoned generator would leave the pool's shutdown waiting): stage this frame unpinned
                return torch.from_numpy(rec), None
        buf, ev = slot
        if tuple(buf.shape) != tuple(rec.shape):  # a frame of another size: stage it unpinned
            free.put(slot)
            return torch.from_numpy(rec), None
        ev.synchronize()  # the copy that last read this buffer is done
        np.copyto(buf.numpy(), rec)
        return buf, slot

    with ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
        pending = []
        nxt = 0

        def submit_batch():
            nonlocal nxt
            if nxt < n:
                idx = list(range(nxt, min(n, nxt + batch_size)))
                pending.append([pool.submit(stage, i) for i in idx])
                nxt += len(idx)

        for _ in range(ahead):
            submit_batch()
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        while pending:
            futs = pending.pop(0)
            submit_batch()
            items = []
            for f in futs:  # device work stays on the caller's thread / stream
                host, slot = f.result()
                rec = host.to(dev, non_blocking=True)
                if slot is not None:
                    slot[1].record()
                    free.put(slot)
                items.append(ds.transform(rec))
            if include_labels:
                yield {'image': torch.stack([it['image'] for it in items]), 'label': torch.stack([it['label'] for it in items])}
            else:
                yield torch.stack(items)
