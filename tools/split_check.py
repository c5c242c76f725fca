#!/usr/bin/env python
"""f32 MFMA vs 3xbf16-split MFMA on the same f32 tensors: error against an f64 torch conv, and time.
   split_check.py            -> a few DeepLab shapes (fwd, dgrad, wgrad)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
import ctypes
import torch
import torch.nn.functional as F
from dass_hip import ops
from dass_hip._lib import lib, check

SHAPES = [  # n h w c k ks stride pad dil
    (8, 129, 129, 304, 256, 3, 1, 1, 1),
    (8, 33, 33, 256, 256, 3, 1, 2, 2),
    (8, 33, 33, 1024, 256, 1, 1, 0, 1),
    (8, 33, 33, 256, 1024, 1, 1, 0, 1),
    (8, 129, 129, 64, 64, 3, 1, 1, 1),
    (8, 65, 65, 128, 128, 3, 2, 1, 1),
    (2, 33, 33, 2048, 256, 3, 1, 12, 12),
]


def timeit(f, reps=10):
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def rel(a, b):
    return ((a.double() - b).norm() / b.norm()).item(), ((a.double() - b).abs().max() / b.abs().max()).item()


for (n, h, w, c, k, ks, st, pad, dil) in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(0)
    oh, ow = ops.conv_out_size(h, ks, st, pad, dil), ops.conv_out_size(w, ks, st, pad, dil)
    x = torch.randn((n, h, w, c), device="cuda", generator=g)
    wt = torch.randn((k, ks, ks, c), device="cuda", generator=g) * (2.0 / (ks * ks * c)) ** 0.5
    dy = torch.randn((n, oh, ow, k), device="cuda", generator=g)
    flops = 2.0 * n * oh * ow * k * ks * ks * c
    x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = wt.double().permute(0, 3, 1, 2).requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, st, pad, dil)
    y64.backward(dy.double().permute(0, 3, 1, 2))
    ref_y = y64.detach().permute(0, 2, 3, 1); ref_dw = w64.grad.permute(0, 2, 3, 1)
    line = "M=%6d C=%4d K=%4d k%d s%d d%-2d" % (n * oh * ow, c, k, ks, st, dil)
    for mode in ("f32", "bf16x3", "bf16x6"):
        ops.set_f32_mma(mode)
        y = torch.empty((n, oh, ow, k), device="cuda")
        wop = ops.prepare_conv_weight(wt)
        f = lambda: ops.conv_launch(x, c, wop, y, k, (n, h, w, c, oh, ow, k, ks, ks, st, pad, dil))
        t = timeit(f)
        e = rel(y, ref_y)
        dw = torch.empty((k, ks, ks, c), device="cuda")
        fw = lambda: check(lib.dass_conv2d_wgrad(ops._p(x), c, ops._p(dy), k, ops._p(dw), n, h, w, c, oh, ow, k, ks, ks, st, pad, dil,
                                                 ops._cdt(dy), ops._stream()), "wgrad")
        tw = timeit(fw)
        ew = rel(dw, ref_dw)
        line += "\n   %6s fwd %.3f ms %6.1f TF/s err %.1e/%.1e  wgrad %.3f ms %6.1f TF/s err %.1e/%.1e |" % (
            mode, t, flops / t / 1e9, e[0], e[1], tw, flops / tw / 1e9, ew[0], ew[1])
    ops.set_f32_mma("f32")
    print(line, flush=True)
