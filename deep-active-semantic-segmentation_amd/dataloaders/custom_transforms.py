"""Geometry and resampling tables of the pool reader -- host side of csrc/pool_reader.hip.

Mirrors what dataloaders/custom_transforms.py:138-166 (FixScaleCrop), :277-297 (FixScaleCropImageOnly) and :214-275
(ScaleWithPadding[ImageOnly]) do to one image, with the pixels left to the device: this module only decides sizes and
offsets and builds the integer tables that make the kernels reproduce scipy.misc.imresize (= PIL's resampler: two-pass
triangle filter in 22-bit fixed point for 'bilinear', a running-sum index walk for 'nearest') bit for bit.  Tables depend
on (input size, output size) only and are cached: a Cityscapes pool builds them once.
"""
import functools
import math

import numpy as np

PRECISION_BITS = 22


@functools.lru_cache(maxsize=64)
def resample_tables(in_size, out_size):
    """-> (first[out], count[out], taps[out, ksize]) int32 numpy arrays for one axis of the bilinear resize"""
    scale = float(in_size) / out_size
    fscale = scale if scale > 1.0 else 1.0
    support = fscale  # triangle filter: support 1, widened when shrinking
    ksize = int(math.ceil(support)) * 2 + 1
    centers = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    first = np.maximum((centers - support + 0.5).astype(np.int64), 0)          # C (int) cast truncates; arguments are >= -0.5 + eps
    first = np.where(centers - support + 0.5 < 0, 0, first)
    last = np.minimum((centers + support + 0.5).astype(np.int64), in_size)
    count = (last - first).astype(np.int64)
    j = np.arange(ksize, dtype=np.float64)[None, :]
    w = 1.0 - np.abs((j + first[:, None] - centers[:, None] + 0.5) * (1.0 / fscale))
    w = np.where((w > 0) & (j < count[:, None]), w, 0.0)
    tot = np.zeros(out_size, dtype=np.float64)
    for col in range(ksize):  # left-to-right accumulation, as the C loop does (the order decides the last bit of the sum)
        tot = tot + w[:, col]
    w = np.where(tot[:, None] != 0.0, w / np.where(tot == 0.0, 1.0, tot)[:, None], w)
    taps = (0.5 + w * float(1 << PRECISION_BITS)).astype(np.int64)             # weights are non-negative for this filter
    taps = np.where(j < count[:, None], taps, 0)
    return first.astype(np.int32), count.astype(np.int32), np.ascontiguousarray(taps.astype(np.int32))


@functools.lru_cache(maxsize=64)
def nearest_table(in_size, out_size):
    """source index per output pixel of imresize(..., 'nearest'): PIL walks xo = scale / 2, xo += scale in double"""
    scale = float(in_size) / out_size
    idx = np.empty(out_size, dtype=np.int32)
    xo = scale * 0.5
    for x in range(out_size):
        idx[x] = int(xo)
        xo += scale
    return np.minimum(idx, in_size - 1).astype(np.int32)


def fix_scale_crop(h, w, crop_size):
    """FixScaleCrop geometry: (resized h, resized w, crop origin y, crop origin x)"""
    if w > h:
        oh = crop_size
        ow = int(1.0 * w * oh / h)
    else:
        ow = crop_size
        oh = int(1.0 * h * ow / w)
    return oh, ow, int(round((oh - crop_size) / 2.)), int(round((ow - crop_size) / 2.))


def scale_with_padding(h, w, base_size=512):
    """ScaleWithPadding geometry: (resized h, resized w, paste origin y, paste origin x) on the base_size canvas"""
    if w < h:
        oh = base_size
        ow = int(1.0 * w * oh / h)
        if ow % 2 != 0:
            ow += 1
    else:
        ow = base_size
        oh = int(1.0 * h * ow / w)
        if oh % 2 != 0:
            oh += 1
    return oh, ow, base_size // 2 - oh // 2, base_size // 2 - ow // 2
