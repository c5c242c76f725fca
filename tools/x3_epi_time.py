#!/usr/bin/env python
"""What the fused epilogues cost: the same conv launched plain (dass_conv2d_x3), with the forward BN statistics (f64 atomics,
dass_conv2d_x3_sums) and as an input gradient carrying the previous layer's BN-backward sums (dass_conv2d_x3_dgrad_bnstats), on
the layer-3 shapes of R101.   DASS_F32_MMA=f16x3 python tools/x3_epi_time.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from dass_hip import ops  # noqa: E402
from dass_hip._lib import check, lib  # noqa: E402
from conv_sweep import timeit  # noqa: E402

ops.set_f32_mma("f16x3")
dev = "cuda"
for name, n, h, c, k, ks in [("l3.c1", 8, 33, 1024, 256, 1), ("l3.c2", 8, 33, 256, 256, 3), ("l3.c3", 8, 33, 256, 1024, 1), ("l2.c3", 8, 65, 128, 512, 1),
                             ("l1.c3", 8, 129, 64, 256, 1)]:
    pad = ks // 2
    m = n * h * h
    x = torch.randn((n, h, h, c), device=dev)
    wt = torch.randn((k, ks, ks, c), device=dev) * 0.05
    y = torch.empty((n, h, h, k), device=dev)
    w3 = ops.prepare_conv_weight(wt, x3=True)
    x3 = ops.split3_rows(x, c, m, c)
    dims = (n, h, h, c, h, h, k, ks, ks, 1, pad, 1)
    ws = ops._x3_workspace(x.device)
    t_plain = timeit(lambda: ops.conv_x3_launch(x3, w3, y, k, dims)) * 1e3
    sums = torch.zeros((2, k), dtype=torch.float64, device=dev)
    t_sums = timeit(lambda: check(lib.dass_conv2d_x3_sums(ops._p(x3), ops._p(w3), ops._p(y), k, n, h, h, c, h, h, k, ks, ks, 1, pad, 1, ops._p(sums),
                                                          ops._p(ws), ws.numel(), ops._stream()), "sums")) * 1e3
    # as an input gradient whose output (K channels) is the d_out of a conv + BN + ReLU layer with conv output yl
    yl = torch.randn((n, h, h, k), device=dev)
    mean = torch.zeros((k,), device=dev)
    invstd = torch.ones((k,), device=dev)
    gsc, gsh = torch.ones((k,), device=dev), torch.zeros((k,), device=dev)
    bsums = torch.zeros((2 * k + k,), dtype=torch.float64, device=dev)
    fused = ctypes.c_int(0)
    res = torch.randn((n, h, h, k), device=dev)

    def bn(residual):
        check(lib.dass_conv2d_x3_dgrad_bnstats(ops._p(x3), ops._p(w3), ops._p(y), k, ops._p(residual), k if residual is not None else 0, n, h, h, c, h, h, k,
                                               ks, ks, pad, 1, ops._p(yl), ops._p(mean), ops._p(invstd), ops._p(gsc), ops._p(gsh), None, 0, 1, ops._p(bsums),
                                               ctypes.byref(fused), ops._p(ws), ws.numel(), ops._stream()), "bnstats")

    t_bn = timeit(lambda: bn(None)) * 1e3
    t_bnr = timeit(lambda: bn(res)) * 1e3
    t_res = timeit(lambda: ops.conv_x3_launch(x3, w3, y, k, dims, residual=res, ldr=k)) * 1e3
    mb = m * k * 4 / 1e6
    print("%-6s M %6d C %4d K %4d ks %d | plain %6.1f us | +fwd sums %6.1f | +residual %6.1f | dgrad+bnstats %6.1f (fused %d) | +residual %6.1f | output %.1f MB"
          % (name, m, c, k, ks, t_plain, t_sums, t_res, t_bn, fused.value, t_bnr, mb))
