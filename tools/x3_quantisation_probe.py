#!/usr/bin/env python
"""What the 3 / 2 tile imbalance costs the layer-3 shapes: the same convs at M = 8 x 33 x 33 = 8712 rows (548 tiles of 64 x 64 on 256 CUs: 3 on 36 CUs, 2 on
the rest) and at M = 8 x 32 x 32 = 8192 rows (512 tiles: exactly 2 per CU) and M = 12 x 32 x 32 = 12288 (768 tiles: exactly 3), per row.  python tools/x3_quantisation_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from dass_hip import ops  # noqa: E402
from conv_sweep import timeit  # noqa: E402

ops.set_f32_mma("f16x3")
for name, c, k, ks in (("l3.c1", 1024, 256, 1), ("l3.c2", 256, 256, 3), ("l3.c3", 256, 1024, 1)):
    out = []
    for n, h in ((8, 33), (8, 32), (12, 32)):
        m = n * h * h
        x = torch.randn((n, h, h, c), device="cuda")
        wt = torch.randn((k, ks, ks, c), device="cuda") * 0.05
        y = torch.empty((n, h, h, k), device="cuda")
        w3 = ops.prepare_conv_weight(wt, x3=True)
        x3 = ops.split3_rows(x, c, m, c)
        dims = (n, h, h, c, h, h, k, ks, ks, 1, ks // 2, 1)
        t = timeit(lambda: ops.conv_x3_launch(x3, w3, y, k, dims)) * 1e3
        tiles = ((m + 63) // 64) * ((k + 63) // 64)
        out.append("M %5d (%4d tiles, %.2f per CU): %5.1f us = %.2f ns per row" % (m, tiles, tiles / 256.0, t, 1e3 * t / m))
    print("%s  %s" % (name, " | ".join(out)), flush=True)
