#!/usr/bin/env python
"""time one conv shape: conv_time.py N H W C K ks stride pad dil  -> ms, TFLOP/s"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
import torch
from dass_hip import ops
n, h, w, c, k, ks, st, pad, dil = [int(v) for v in sys.argv[1:10]]
oh, ow = ops.conv_out_size(h, ks, st, pad, dil), ops.conv_out_size(w, ks, st, pad, dil)
x = torch.randn((n, h, w, c), device="cuda"); wt = torch.randn((k, ks, ks, c), device="cuda") * 0.05
y = torch.empty((n, oh, ow, k), device="cuda")
wop = ops.prepare_conv_weight(wt)
f = lambda: ops.conv_launch(x, c, wop, y, k, (n, h, w, c, oh, ow, k, ks, ks, st, pad, dil))
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("M=%d C=%d K=%d k%d: %.3f ms  %.1f TFLOP/s" % (n * oh * ow, c, k, ks, ms, 2.0 * n * oh * ow * k * ks * ks * c / ms / 1e9))
