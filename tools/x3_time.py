#!/usr/bin/env python
"""Times the pipelined pre-split conv engine (dass_conv2d_x3) against the classic bf16x6 kernel on every distinct
DeepLab-R101 shape (forward form; dgrad is the same kernel with the roles of C and K swapped), per forced tile and with
the cost model's own choice, plus the f32 -> x3 conversion pass.  GPU only.   python tools/x3_time.py [tiles...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from dass_hip import ops  # noqa: E402
from dass_hip._lib import lib  # noqa: E402
from conv_sweep import r101_shapes, timeit  # noqa: E402


def main():
    tiles = [int(a) for a in sys.argv[1:]] or [0, 11, 12, 14, 21, 22, 23, 24]   # + 1000: no loader wave, + 2000: loader wave (whole-tile launches)
    ops.set_f32_mma(os.environ.get("DASS_F32_MMA", "bf16x6"))   # (f16x3: the same kernels in their two-part mode)
    dev = "cuda"
    tot_old, tot_new, tot_best, tot_flop, tot_split = 0.0, {t: 0.0 for t in tiles}, 0.0, 0.0, 0.0
    print("%-14s %3s %7s %5s %5s | %8s | %s | %7s" % ("shape", "cnt", "M", "C", "K", "old us", " ".join("t%-2d us " % t for t in tiles), "split us"))
    for name, cnt, n, h, w, c, k, ks, st, pad, dil in r101_shapes():
        if c < 16 or k < 32:
            continue
        oh, ow = ops.conv_out_size(h, ks, st, pad, dil), ops.conv_out_size(w, ks, st, pad, dil)
        x = torch.randn((n, h, w, c), device=dev)
        wt = torch.randn((k, ks, ks, c), device=dev) * 0.05
        y = torch.empty((n, oh, ow, k), device=dev)
        dims = (n, h, w, c, oh, ow, k, ks, ks, st, pad, dil)
        w3 = ops.prepare_conv_weight(wt, x3=True)
        w6 = ops.prepare_conv_weight(wt)
        x3 = ops.split3_rows(x, c, n * h * w, c)
        t_old = timeit(lambda: ops.conv_launch(x, c, w6, y, k, dims)) * 1e3
        t_split = timeit(lambda: ops.split3_rows(x, c, n * h * w, c)) * 1e3
        res = {}
        for t in tiles:
            lib.dass_x3_force_tile(t)
            res[t] = timeit(lambda: ops.conv_x3_launch(x3, w3, y, k, dims)) * 1e3
        lib.dass_x3_force_tile(0)
        flop = 2.0 * n * oh * ow * k * ks * ks * c
        best = min(res.values())
        print("%-14s %3d %7d %5d %5d | %8.1f | %s | %7.1f   best %.0f TF/s (old %.0f)" % (
            name, cnt, n * oh * ow, c, k, t_old, " ".join("%7.1f" % res[t] for t in tiles), t_split, flop / best / 1e6, flop / t_old / 1e6))
        tot_old += cnt * t_old
        tot_best += cnt * best
        tot_flop += cnt * flop
        tot_split += cnt * t_split
        for t in tiles:
            tot_new[t] += cnt * res[t]
    print("count-weighted forward totals: old %.2f ms | %s | best-per-shape %.2f ms | split pass %.2f ms" % (
        tot_old / 1e3, " ".join("t%d %.2f" % (t, tot_new[t] / 1e3) for t in tiles), tot_best / 1e3, tot_split / 1e3))
    print("aggregate: old %.1f TF/s, best-per-shape %.1f TF/s of 416.7" % (tot_flop / tot_old / 1e6, tot_flop / tot_best / 1e6))


if __name__ == "__main__":
    main()
