#!/usr/bin/env python
"""Hot vs cold operand: the forward conv of the layer-3 shapes launched 16 times over ONE input buffer (resident in the Infinity Cache
after the first launch: what tools/x3_time.py measures) and over a ROTATION of buffers totalling > 600 MB (every launch reads its input
from HBM, as in the train step, where the producer's output has long left the caches... or has it).   python tools/x3_cold_time.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
import torch  # noqa: E402
from dass_hip import ops  # noqa: E402

ops.set_f32_mma("f16x3")
dev = "cuda"
for name, n, h, c, k, ks in [("l3.c1", 8, 33, 1024, 256, 1), ("l3.c2", 8, 33, 256, 256, 3), ("l3.c3", 8, 33, 256, 1024, 1), ("l2.c1", 8, 65, 512, 128, 1),
                             ("l1.c3", 8, 129, 64, 256, 1)]:
    pad = ks // 2
    m = n * h * h
    wt = torch.randn((k, ks, ks, c), device=dev) * 0.05
    w3 = ops.prepare_conv_weight(wt, x3=True)
    dims = (n, h, h, c, h, h, k, ks, ks, 1, pad, 1)
    in_mb = m * c * 4 / 1e6
    nbuf = max(2, int(700 / max(in_mb + m * k * 4 / 1e6, 1)) + 1)
    xs = [ops.split3_rows(torch.randn((n, h, h, c), device=dev), c, m, c) for _ in range(nbuf)]
    ys = [torch.empty((n, h, h, k), device=dev) for _ in range(nbuf)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def run(rot):
        reps = 4 * nbuf
        for i in range(nbuf):
            ops.conv_x3_launch(xs[i if rot else 0], w3, ys[i if rot else 0], k, dims)
        torch.cuda.synchronize()
        e0.record()
        for i in range(reps):
            j = i % nbuf if rot else 0
            ops.conv_x3_launch(xs[j], w3, ys[j], k, dims)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    hot, cold = run(False), run(True)
    print("%-6s M %6d C %4d K %4d ks %d | input %5.1f MB output %5.1f MB | hot %6.1f us | cold (%d buffers) %6.1f us" % (name, m, c, k, ks, in_mb, m * k * 4 / 1e6, hot, nbuf, cold))
