#!/usr/bin/env python
"""time a few representative DeepLab shapes (forward conv only) in the current engine / tile knobs"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
import torch
from dass_hip import ops
SHAPES = [("dec.3x3a", 8, 129, 304, 256, 3, 1, 1), ("l3.c2", 8, 33, 256, 256, 3, 1, 1), ("l3.c1", 8, 33, 1024, 256, 1, 0, 1), ("l3.c3", 8, 33, 256, 1024, 1, 0, 1),
          ("l4.c2.d4", 8, 33, 512, 512, 3, 4, 4), ("aspp.d12", 8, 33, 2048, 256, 3, 12, 12), ("l2.c2", 8, 65, 128, 128, 3, 1, 1), ("l1.c2", 8, 129, 64, 64, 3, 1, 1),
          ("l2.c3", 8, 65, 128, 512, 1, 0, 1)]
out = []
for name, n, h, c, k, ks, pad, dil in SHAPES:
    x = torch.randn((n, h, h, c), device="cuda"); wt = torch.randn((k, ks, ks, c), device="cuda") * 0.05
    y = torch.empty((n, h, h, k), device="cuda"); wop = ops.prepare_conv_weight(wt)
    f = lambda: ops.conv_launch(x, c, wop, y, k, (n, h, h, c, h, h, k, ks, ks, 1, pad, dil))
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    out.append("%s %.3f (%.0f)" % (name, ms, 2.0 * n * h * h * k * ks * ks * c / ms / 1e9))
print("%s tile=%s pf=%s | %s" % (ops.f32_mma(), os.environ.get("DASS_CONV_TILE", "auto"), os.environ.get("DASS_CONV_PF", "-"), " | ".join(out)), flush=True)
