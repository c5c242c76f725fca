#!/usr/bin/env python
"""bandwidth of the BN elementwise / reduction kernels on DeepLab-sized tensors: GB/s per kernel and shape"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
import torch
from dass_hip import ops
from dass_hip._lib import lib, check


def timeit(f, reps=20):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for m, k in ((8712, 1024), (8712, 256), (33800, 512), (133128, 256), (133128, 64)):
    x = torch.randn((m, k), device="cuda"); out = torch.empty_like(x); dy = torch.randn_like(x); dx = torch.empty_like(x)
    sc = torch.rand(k, device="cuda") + 0.5; sh = torch.randn(k, device="cuda")
    mean = torch.zeros(k, device="cuda"); inv = torch.ones(k, device="cuda"); db = torch.randn(k, device="cuda"); dg = torch.randn(k, device="cuda")
    t1 = timeit(lambda: ops.scale_shift_act(x, k, out, k, m, k, sc, sh, act=ops.ACT_RELU))
    nrows = lib.dass_stat_rows(m); partial = torch.empty((nrows, 2, k), device="cuda")
    t2 = timeit(lambda: check(lib.dass_bn_bwd_reduce(ops._p(dy), k, ops._p(out), k, ops._p(x), k, ops._p(mean), ops._p(inv), None, m, k, 1, ops.ACT_RELU,
                                                     ops._p(partial), 0, ops._stream()), "r"))
    t3 = timeit(lambda: check(lib.dass_bn_bwd_apply(ops._p(dy), k, ops._p(out), k, ops._p(x), k, ops._p(mean), ops._p(inv), ops._p(sc), ops._p(db), ops._p(dg), None,
                                                    ops._p(dx), k, None, 0, m, k, 1, float(m), 1, ops.ACT_RELU, 0, None, ops._stream()), "a"))
    b = m * k * 4 / 1e9
    print("M=%6d K=%4d (%.1f MB): scale_shift_act %.1f us %.0f GB/s | bwd_reduce %.1f us %.0f GB/s | bwd_apply %.1f us %.0f GB/s"
          % (m, k, b * 1e3, t1 * 1e3, 2 * b / t1 * 1e3, t2 * 1e3, 3 * b / t2 * 1e3, t3 * 1e3, 4 * b / t3 * 1e3), flush=True)
