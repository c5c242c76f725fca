"""CPU oracle of the pool reader: LMDB record -> resized / cropped / normalised tensors.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates
  dataloaders/dataset/paths_dataset.py:27-52           record = pickle(np.uint8[H, W, 4]) = RGB + label, transform chains
  dataloaders/custom_transforms.py:8-51                Normalize (/255, -mean, /std in float32), ToTensor (HWC -> CHW float)
  dataloaders/custom_transforms.py:138-166,277-297     FixScaleCrop / FixScaleCropImageOnly (short side -> crop, centre crop)
  dataloaders/custom_transforms.py:214-245             ScaleWithPadding (crop_size == -1: long side -> 512, centred on a 512^2 canvas)
The resize itself is third-party: the reference calls scipy.misc.imresize (removed from SciPy >= 1.3, un-pinned by the
reference: no requirements file), whose published implementation is `toimage(arr).resize((w, h), resample=BILINEAR |
NEAREST)` on a uint8 PIL image.  PIL's resampler is restated here in integer numpy -- separable two-pass (horizontal, then
vertical) convolution with the triangle filter widened by the down-scale factor, coefficients normalised and rounded to 22
fractional bits, every pass rounded to uint8 (Pillow src/libImaging/Resample.c) -- and pinned against Pillow itself in
tests/test_cpu.py and, through the reference's own transform classes, in oracle/make_goldens_r2.py.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def resample_coeffs(in_size, out_size):
    """-> (xmin[out], count[out], kk[out, ksize] int32): Pillow's precompute_coeffs + normalize_coeffs_8bpc, bilinear"""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, dtype=np.int32)
    cnt = np.zeros(out_size, dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        w = np.array([max(0.0, 1.0 - abs((x + lo - center + 0.5) * ss)) for x in range(n)], dtype=np.float64)
        tot = 0.0
        for v in w:  # the C loop accumulates in order
            tot += v
        if tot != 0.0:
            w = w / tot
        fixed = np.where(w < 0, (-0.5 + w * (1 << PRECISION_BITS)).astype(np.int64), (0.5 + w * (1 << PRECISION_BITS)).astype(np.int64))
        xmin[xx], cnt[xx] = lo, n
        kk[xx, :n] = fixed.astype(np.int32)
    return xmin, cnt, kk


def _pass(arr, out_size, axis):
    """one resampling pass of a uint8 array along `axis` (0 = vertical, 1 = horizontal)"""
    a = np.moveaxis(arr, axis, 0).astype(np.int64)
    xmin, cnt, kk = resample_coeffs(a.shape[0], out_size)
    out = np.empty((out_size,) + a.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        acc = np.full(a.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for i in range(cnt[xx]):
            acc += a[xmin[xx] + i] * int(kk[xx, i])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(arr, oh, ow):
    """scipy.misc.imresize(arr, (oh, ow)) for uint8 input: horizontal pass first, then vertical (Pillow ImagingResample)"""
    out = arr
    if out.shape[1] != ow:
        out = _pass(out, ow, 1)
    if out.shape[0] != oh:
        out = _pass(out, oh, 0)
    return out


def nearest_indices(in_size, out_size):
    """source index of every output pixel: Pillow resizes with NEAREST through its scale-only affine path
    (src/libImaging/Geometry.c ImagingScaleAffine): xo = scale / 2, then xo += scale per pixel IN DOUBLE (the running sum,
    not (x + 0.5) * scale: they differ in the last bit at exact pixel boundaries), index = (int) xo"""
    scale = float(in_size) / out_size
    idx = np.empty(out_size, dtype=np.int64)
    xo = scale * 0.5
    for x in range(out_size):
        idx[x] = int(xo)
        xo += scale
    return np.minimum(idx, in_size - 1)


def resize_nearest_u8(arr, oh, ow):
    """scipy.misc.imresize(arr, (oh, ow), 'nearest')"""
    h, w = arr.shape[:2]
    return arr[nearest_indices(h, oh)][:, nearest_indices(w, ow)]


def fix_scale_crop_geometry(h, w, crop):
    """custom_transforms.py:144-160: short side -> crop (aspect kept with int()), then the centre crop offsets"""
    if w > h:
        oh, ow = crop, int(1.0 * w * crop / h)
    else:
        ow, oh = crop, int(1.0 * h * crop / w)
    x1, y1 = int(round((ow - crop) / 2.0)), int(round((oh - crop) / 2.0))
    return oh, ow, y1, x1


def pad_scale_geometry(h, w, base=512):
    """custom_transforms.py:225-236: long side -> base, the other side rounded UP to even; centred paste offsets"""
    if w < h:
        oh, ow = base, int(1.0 * w * base / h)
        ow += ow % 2
    else:
        ow, oh = base, int(1.0 * h * base / w)
        oh += oh % 2
    return oh, ow, base // 2 - oh // 2, base // 2 - ow // 2


def normalize_chw(img_hwc_u8_or_f32, divide=True):
    """the LABEL chain: custom Normalize + ToTensor (custom_transforms.py:8-51).  `img /= 255.0` is a float32 op; `img -= mean`
    and `img /= std` take TUPLES of python floats, which numpy turns into float64 arrays: each op runs in double and is
    stored back as float32.  HWC -> CHW"""
    img = img_hwc_u8_or_f32.astype(np.float32)
    if divide:
        img /= 255.0
    img -= np.array(MEAN, dtype=np.float64)
    img /= np.array(STD, dtype=np.float64)
    return np.ascontiguousarray(img.transpose(2, 0, 1))


def normalize_chw_torchvision(img_hwc, divide=True):
    """the IMAGE-ONLY chain: torchvision transforms.ToTensor (uint8 HWC -> float32 CHW / 255; a FLOAT array is passed through
    undivided) + transforms.Normalize (float32 mean / std tensors: (x - mean) / std in float32)"""
    img = np.ascontiguousarray(img_hwc.transpose(2, 0, 1)).astype(np.float32)
    if divide:
        img = img / np.float32(255.0)
    mean, std = np.array(MEAN, dtype=np.float32)[:, None, None], np.array(STD, dtype=np.float32)[:, None, None]
    return (img - mean) / std


def pool_sample(record, crop_size, include_labels):
    """PathsDataset.__getitem__ on one decoded record [H, W, 4] uint8 -> {'image': f32 [3, S, S], 'label': f32 [S, S]} or image"""
    image, target = record[:, :, 0:3], record[:, :, 3]
    h, w = image.shape[:2]
    if crop_size == -1:
        oh, ow, y0, x0 = pad_scale_geometry(h, w)
        canvas = np.zeros((512, 512, 3), dtype=np.float32)
        canvas[y0:y0 + oh, x0:x0 + ow] = resize_bilinear_u8(image, oh, ow)
        if include_labels:
            mask = np.ones((512, 512), dtype=np.uint8) * 255
            mask[y0:y0 + oh, x0:x0 + ow] = resize_nearest_u8(target, oh, ow)
            return {"image": normalize_chw(canvas), "label": mask.astype(np.float32)}
        # ScaleWithPaddingImageOnly hands torchvision's ToTensor a FLOAT array, which it does not divide by 255
        return normalize_chw_torchvision(canvas, divide=False)
    oh, ow, y1, x1 = fix_scale_crop_geometry(h, w, crop_size)
    img = resize_bilinear_u8(image, oh, ow)[y1:y1 + crop_size, x1:x1 + crop_size]
    if include_labels:
        mask = resize_nearest_u8(target, oh, ow)[y1:y1 + crop_size, x1:x1 + crop_size]
        return {"image": normalize_chw(img), "label": mask.astype(np.float32)}
    return normalize_chw_torchvision(img)
