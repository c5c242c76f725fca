#!/usr/bin/env python
"""Per-tile fixed cost of the pre-split conv kernel: 3x3 conv at 8 x 129 x 129 -> 256 channels with C = 64 / 128 / 256 input
channels (18 / 36 / 72 slabs per tile) and three epilogues; a linear fit in the slab count separates the main-loop rate
from what every tile pays for set-up + epilogue.  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from dass_hip import ops  # noqa: E402
from dass_hip._lib import lib  # noqa: E402
from conv_sweep import timeit  # noqa: E402


def main():
    ops.set_f32_mma("bf16x6")
    n, h, k = 8, 129, 256
    m = n * h * h
    res = torch.randn(m, k, device="cuda")
    sc, sh = torch.rand(k, device="cuda") + 0.5, torch.randn(k, device="cuda")
    for code in [int(a) for a in sys.argv[1:]] or [0]:
        lib.dass_x3_force_tile(code)
        print("tile code", code)
        for ep in ("y f32", "y f32 + bn/relu", "y3 + residual + bn/relu"):
            ts = []
            for c in (64, 128, 256):
                x = torch.randn(n, h, h, c, device="cuda")
                w = torch.randn(k, 3, 3, c, device="cuda") * 0.02
                x3, w3 = ops.split3_rows(x, c, m, c), ops.prepare_conv_weight(w)
                dims = (n, h, h, c, h, h, k, 3, 3, 1, 1, 1)
                y = torch.empty(m, k, device="cuda")
                y3 = ops.x3_alloc(m, k, "cuda")
                if ep == "y f32":
                    f = lambda: ops.conv_x3_launch(x3, w3, y, k, dims)
                elif ep == "y f32 + bn/relu":
                    f = lambda: ops.conv_x3_launch(x3, w3, y, k, dims, scale=sc, shift=sh, act=ops.ACT_RELU)
                else:
                    f = lambda: ops.conv_x3_launch(x3, w3, None, 0, dims, y3=y3, scale=sc, shift=sh, residual=res, ldr=k, act=ops.ACT_RELU)
                ts.append(timeit(f) * 1e3)
            per_slab = (ts[2] - ts[0]) / 54.0
            print("  %-26s C=64 %7.1f us  C=128 %7.1f us  C=256 %7.1f us | %.2f us per slab-of-all-tiles, intercept %.1f us (%.0f%% of C=256), loop rate %.0f TF/s"
                  % (ep, ts[0], ts[1], ts[2], per_slab, ts[2] - 72 * per_slab, 100 * (ts[2] - 72 * per_slab) / ts[2],
                     2.0 * m * k * 9 * 256 / (72 * per_slab) / 1e6), flush=True)


if __name__ == "__main__":
    main()
