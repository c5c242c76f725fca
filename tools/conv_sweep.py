#!/usr/bin/env python
"""Times every distinct conv shape of DeepLab-R101 (os16, 513^2, batch 8) through the C-ABI: forward,
dgrad and wgrad, with events on the launch stream.  Prints one row per (shape, pass) and the
count-weighted totals -- the worklist for kernel tuning.  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
import torch  # noqa: E402
from dass_hip import ops  # noqa: E402
from dass_hip._lib import check, lib  # noqa: E402

PEAK = {"f32": 157.3, "bf16x6": 2500.0 / 6, "bf16x3": 2500.0 / 3, "f16x3": 2500.0 / 3}[ops.f32_mma()]  # engine from DASS_F32_MMA


def r101_shapes(batch=8, size=513):
    """(name, count, N, H, W, C, K, ksize, stride, pad, dil)"""
    s2 = (size + 1) // 2      # 257
    s4 = (s2 + 1) // 2        # 129
    s8 = (s4 + 1) // 2        # 65
    s16 = (s8 + 1) // 2       # 33
    L = []
    L.append(("stem7x7", 1, batch, size, size, 4, 64, 7, 2, 3, 1))
    # layer1 @129
    L += [("l1.c1.first", 1, batch, s4, s4, 64, 64, 1, 1, 0, 1), ("l1.c1", 2, batch, s4, s4, 256, 64, 1, 1, 0, 1),
          ("l1.c2", 3, batch, s4, s4, 64, 64, 3, 1, 1, 1), ("l1.c3", 3, batch, s4, s4, 64, 256, 1, 1, 0, 1),
          ("l1.down", 1, batch, s4, s4, 64, 256, 1, 1, 0, 1)]
    # layer2 -> 65
    L += [("l2.c1.first", 1, batch, s4, s4, 256, 128, 1, 1, 0, 1), ("l2.c2.s2", 1, batch, s4, s4, 128, 128, 3, 2, 1, 1),
          ("l2.down.s2", 1, batch, s4, s4, 256, 512, 1, 2, 0, 1),
          ("l2.c1", 3, batch, s8, s8, 512, 128, 1, 1, 0, 1), ("l2.c2", 3, batch, s8, s8, 128, 128, 3, 1, 1, 1),
          ("l2.c3", 4, batch, s8, s8, 128, 512, 1, 1, 0, 1)]
    # layer3 -> 33
    L += [("l3.c1.first", 1, batch, s8, s8, 512, 256, 1, 1, 0, 1), ("l3.c2.s2", 1, batch, s8, s8, 256, 256, 3, 2, 1, 1),
          ("l3.down.s2", 1, batch, s8, s8, 512, 1024, 1, 2, 0, 1),
          ("l3.c1", 22, batch, s16, s16, 1024, 256, 1, 1, 0, 1), ("l3.c2", 22, batch, s16, s16, 256, 256, 3, 1, 1, 1),
          ("l3.c3", 23, batch, s16, s16, 256, 1024, 1, 1, 0, 1)]
    # layer4 @33 dil 2,4,8
    L += [("l4.c1.first", 1, batch, s16, s16, 1024, 512, 1, 1, 0, 1), ("l4.c1", 2, batch, s16, s16, 2048, 512, 1, 1, 0, 1),
          ("l4.c2.d2", 1, batch, s16, s16, 512, 512, 3, 1, 2, 2), ("l4.c2.d4", 1, batch, s16, s16, 512, 512, 3, 1, 4, 4),
          ("l4.c2.d8", 1, batch, s16, s16, 512, 512, 3, 1, 8, 8), ("l4.c3", 3, batch, s16, s16, 512, 2048, 1, 1, 0, 1),
          ("l4.down", 1, batch, s16, s16, 1024, 2048, 1, 1, 0, 1)]
    # ASPP
    L += [("aspp1", 1, batch, s16, s16, 2048, 256, 1, 1, 0, 1), ("aspp.d6", 1, batch, s16, s16, 2048, 256, 3, 1, 6, 6),
          ("aspp.d12", 1, batch, s16, s16, 2048, 256, 3, 1, 12, 12), ("aspp.d18", 1, batch, s16, s16, 2048, 256, 3, 1, 18, 18),
          ("aspp.merge", 1, batch, s16, s16, 1280, 256, 1, 1, 0, 1)]
    # decoder
    L += [("dec.low", 1, batch, s4, s4, 256, 48, 1, 1, 0, 1), ("dec.3x3a", 1, batch, s4, s4, 304, 256, 3, 1, 1, 1),
          ("dec.3x3b", 1, batch, s4, s4, 256, 256, 3, 1, 1, 1), ("dec.cls", 1, batch, s4, s4, 256, 20, 1, 1, 0, 1)]
    return L


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = "cuda"
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    totf = 0.0
    rows = []
    for name, cnt, n, h, w, c, k, ks, st, pad, dil in r101_shapes():
        oh, ow = ops.conv_out_size(h, ks, st, pad, dil), ops.conv_out_size(w, ks, st, pad, dil)
        x = torch.randn((n, h, w, c), device=dev)
        wt = torch.randn((k, ks, ks, c), device=dev) * 0.05
        wt_t = torch.randn((c, ks, ks, k), device=dev) * 0.05
        y = torch.empty((n, oh, ow, k), device=dev)
        dy = torch.randn((n, oh, ow, k), device=dev)
        dx = torch.empty((n, h, w, c), device=dev)
        dw = torch.empty((k, ks, ks, c), device=dev)
        flops = 2.0 * n * oh * ow * k * ks * ks * c
        stream = ops._stream()
        wop, wop_t = ops.prepare_conv_weight(wt), ops.prepare_conv_weight(wt_t)
        f_fwd = lambda: ops.conv_launch(x, c, wop, y, k, (n, h, w, c, oh, ow, k, ks, ks, st, pad, dil))  # noqa: E731
        pad_t = dil * (ks - 1) - pad
        f_dg = lambda: check(lib.dass_conv2d_igemm(ops._p(dy), k, ops._p(wop_t), ops._p(dx), c, None, None, None, 0, None, n, oh, ow, k,  # noqa: E731
                                                   h, w, c, ks, ks, 1, pad_t, dil, st, 0, ops._cdt(dx), stream), "dgrad")
        f_wg = lambda: check(lib.dass_conv2d_wgrad(ops._p(x), c, ops._p(dy), k, ops._p(dw), n, h, w, c, oh, ow, k, ks, ks, st, pad,  # noqa: E731
                                                   dil, ops._cdt(dy), stream), "wgrad")
        res = {}
        for tag, f in (("fwd", f_fwd), ("dgrad", f_dg), ("wgrad", f_wg)):
            if tag == "dgrad" and name == "stem7x7":
                continue
            ms = timeit(f)
            res[tag] = ms
            tot[tag] += ms * cnt
        totf += flops * cnt
        rows.append((name, cnt, n * oh * ow, c, k, ks, st, dil, flops / 1e9, res))
    print("engine %s  DASS_CONV_TILE=%s" % (ops.f32_mma(), os.environ.get("DASS_CONV_TILE", "auto")))
    print("%-14s %3s %7s %5s %5s k s d %8s | %8s %6s | %8s %6s | %8s %6s" % ("layer", "cnt", "M", "C", "K", "GFLOP", "fwd ms", "TF/s", "dgrad ms", "TF/s", "wgrad ms", "TF/s"))
    for name, cnt, m, c, k, ks, st, dil, gf, res in rows:
        cells = []
        for tag in ("fwd", "dgrad", "wgrad"):
            if tag in res:
                cells.append("%8.3f %6.1f" % (res[tag], gf / res[tag]))
            else:
                cells.append("%8s %6s" % ("-", "-"))
        print("%-14s %3d %7d %5d %5d %d %d %d %8.2f | %s" % (name, cnt, m, c, k, ks, st, dil, gf, " | ".join(cells)))
    print("count-weighted totals per step: fwd %.2f ms, dgrad %.2f ms, wgrad %.2f ms; conv GFLOP/pass %.1f -> at %.1f TF/s peak: %.2f ms"
          % (tot["fwd"], tot["dgrad"], tot["wgrad"], totf / 1e9, PEAK, totf / 1e9 / PEAK))


if __name__ == "__main__":
    main()
