#!/usr/bin/env python
"""Times the pre-split weight-gradient kernel (dass_conv2d_wgrad_x3) against the classic bf16x6 wgrad on every distinct
DeepLab-R101 shape, for both of its tiles (DASS_WX3_TILE is read once per process: run once per tile).  GPU only.
    DASS_WX3_TILE=1 python tools/wx3_time.py ; DASS_WX3_TILE=2 python tools/wx3_time.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from dass_hip import ops  # noqa: E402
from dass_hip._lib import check, lib  # noqa: E402
from conv_sweep import r101_shapes, timeit  # noqa: E402


def main():
    ops.set_f32_mma("bf16x6")
    dev = "cuda"
    tot_old = tot_new = tot_flop = 0.0
    print("DASS_WX3_TILE=%s TARGET=%s MINSLABS=%s" % (os.environ.get("DASS_WX3_TILE", "auto"), os.environ.get("DASS_WX3_TARGET", "768"), os.environ.get("DASS_WX3_MINSLABS", "16")))
    for name, cnt, n, h, w, c, k, ks, st, pad, dil in r101_shapes():
        if c < 16 or k < 32:
            continue
        oh, ow = ops.conv_out_size(h, ks, st, pad, dil), ops.conv_out_size(w, ks, st, pad, dil)
        x = torch.randn((n, h, w, c), device=dev)
        dy = torch.randn((n, oh, ow, k), device=dev)
        dw = torch.empty((k, ks, ks, c), device=dev)
        x3, dy3 = ops.split3_rows(x, c, n * h * w, c), ops.split3_rows(dy, k, n * oh * ow, k)
        st_ = ops._stream()
        t_old = timeit(lambda: check(lib.dass_conv2d_wgrad(ops._p(x), c, ops._p(dy), k, ops._p(dw), n, h, w, c, oh, ow, k, ks, ks, st, pad, dil,
                                                           ops._cdt(dy), st_), "w")) * 1e3
        t_new = timeit(lambda: check(lib.dass_conv2d_wgrad_x3(ops._p(x3), ops._p(dy3), ops._p(dw), n, h, w, c, oh, ow, k, ks, ks, st, pad, dil, 1,
                                                              st_), "w3")) * 1e3
        flop = 2.0 * n * oh * ow * k * ks * ks * c
        print("%-14s %3d M=%7d C=%5d K=%5d k%d | classic %8.1f us %6.1f TF/s | x3 %8.1f us %6.1f TF/s" % (
            name, cnt, n * oh * ow, c, k, ks, t_old, flop / t_old / 1e6, t_new, flop / t_new / 1e6), flush=True)
        tot_old += cnt * t_old
        tot_new += cnt * t_new
        tot_flop += cnt * flop
    print("count-weighted wgrad totals: classic %.2f ms (%.1f TF/s) | x3 %.2f ms (%.1f TF/s)" % (
        tot_old / 1e3, tot_flop / tot_old / 1e6, tot_new / 1e3, tot_flop / tot_new / 1e6))


if __name__ == "__main__":
    main()
