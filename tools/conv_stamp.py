#!/usr/bin/env python
"""per-segment cycle counts of the conv main loop (needs the -DDASS_STAMP debug build: DASS_HIP_LIB=.../libdass_stamp.so).
segments per slab iteration, summed over the loop by wave 0 of 8 sample workgroups:
  0 barrier-1 wait   1 wait for the global loads + LDS write   2 barrier-2   3 issue next loads   4 ds_read + convert + MFMA"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
import torch
from dass_hip import ops
from dass_hip._lib import lib, check
n, h, c, k, ks, pad, dil = [int(v) for v in sys.argv[1:8]]
x = torch.randn((n, h, h, c), device="cuda"); wt = torch.randn((k, ks, ks, c), device="cuda") * 0.05
y = torch.empty((n, h, h, k), device="cuda"); wop = ops.prepare_conv_weight(wt)
partial = torch.zeros((lib.dass_conv2d_igemm_stats_rows(n * h * h) * 2 * k + 64,), device="cuda")
nrows = ctypes.c_int(0)
for _ in range(3):
    check(lib.dass_conv2d_igemm_stats(ops._p(x), c, ops._p(wop), ops._p(y), k, n, h, h, c, h, h, k, ks, ks, 1, pad, dil, ops._cdt(y), ops._p(partial),
                                      ctypes.byref(nrows), ops._stream()), "stats")
torch.cuda.synchronize()
v = partial[:64].view(8, 8).cpu()
nsl = ks * ks * ((c + 31) // 32)
print("shape", sys.argv[1:8], "engine", ops.f32_mma(), "slabs/WG ~", nsl)
for r in v:
    tot = r[0].item()
    print("total %8.0f cyc (%.0f/slab) | " % (tot, tot / nsl) + "  ".join("seg%d %4.1f%%" % (i, 100 * r[1 + i].item() / max(tot, 1)) for i in range(5)))
