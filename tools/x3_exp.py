#!/usr/bin/env python
"""Timing experiments on the pre-split conv kernel (conv_x3.hip X3_EXP builds; results are WRONG by construction): which part
of the slab loop bounds the M = 8712 launches.  python tools/x3_exp.py  (parent: one child per library build)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
WANT = ("l3.c1", "l3.c2", "l3.c3", "l4.c1", "l4.c3", "aspp1", "l2.c1", "l2.c3", "l1.c3")


def child():
    import torch
    from dass_hip import ops
    from dass_hip._lib import lib
    from conv_sweep import r101_shapes, timeit
    ops.set_f32_mma("f16x3")
    out = []
    for name, cnt, n, h, w, c, k, ks, st, pad, dil in r101_shapes():
        if name not in WANT:
            continue
        oh, ow = ops.conv_out_size(h, ks, st, pad, dil), ops.conv_out_size(w, ks, st, pad, dil)
        x = torch.randn((n, h, w, c), device="cuda")
        wt = torch.randn((k, ks, ks, c), device="cuda") * 0.05
        y = torch.empty((n, oh, ow, k), device="cuda")
        dims = (n, h, w, c, oh, ow, k, ks, ks, st, pad, dil)
        w3 = ops.prepare_conv_weight(wt, x3=True)
        x3 = ops.split3_rows(x, c, n * h * w, c)
        lib.dass_x3_force_tile(int(os.environ.get("X3_TILE", "0")))
        if os.environ.get("X3_STAMP") == "1":   # libdass_exp12.so: per-workgroup wall-clock stamps (100 MHz) in the stream-K workspace
            import numpy as np
            ws = ops._x3_workspace(x3.device)
            for _ in range(3):
                ops.conv_x3_launch(x3, w3, y, k, dims)
            ws.zero_()
            torch.cuda.synchronize()
            ops.conv_x3_launch(x3, w3, y, k, dims)
            torch.cuda.synchronize()
            t = ws.view(torch.int64)[: 16 * 16384].view(-1, 16).cpu().numpy()
            t = t[(t[:, 0] > 0) & (t[:, 4] > t[:, 0]) & (t[:, 4] - t[:, 0] < 10 ** 6)]
            us = lambda v: float(v) * 0.01  # noqa: E731
            t0 = t[:, 0].min()
            seg = [np.mean(t[:, i + 1] - t[:, i]) for i in range(4)]
            start = np.percentile(t[:, 0] - t0, [50, 90, 100])
            print("STAMP2 %-8s entry->args %4.2f  ->barrier %4.2f  ->rows %4.2f  ->taps/acc %4.2f" % (name, us(np.mean(t[:, 5] - t[:, 0])), us(np.mean(t[:, 6] - t[:, 5])),
                  us(np.mean(t[:, 7] - t[:, 6])), us(np.mean(t[:, 1] - t[:, 7]))), flush=True)
            w4 = t[:, 8:12].max(axis=1) > 0
            print("STAMP3 %-8s wave entry after wave 0: %s | wave reaches barrier after own entry: %s" % (name,
                  " ".join("%.2f" % us(np.mean(t[w4, 8 + i] - t[w4, 8])) for i in range(4)), " ".join("%.2f" % us(np.mean(t[w4, 12 + i] - t[w4, 8 + i])) for i in range(4))), flush=True)
            print("STAMP %-8s wgs %5d span %6.1f us | prologue %5.1f  first slab %5.1f  loop %5.1f  epilogue %5.1f | lifetime %5.1f | start p50/p90/max %5.1f %5.1f %5.1f | end max %5.1f" % (
                name, len(t), us(t[:, 4].max() - t0), us(seg[0]), us(seg[1]), us(seg[2]), us(seg[3]), us(np.mean(t[:, 4] - t[:, 0])),
                us(start[0]), us(start[1]), us(start[2]), us((t[:, 4] - t0).max())), flush=True)
            continue
        out.append("%s %.1f" % (name, timeit(lambda: ops.conv_x3_launch(x3, w3, y, k, dims)) * 1e3))
    print("RES " + " | ".join(out), flush=True)


if __name__ == "__main__":
    if os.environ.get("X3_CHILD") == "1":
        child()
    else:
        d = os.path.join(ROOT, "deep-active-semantic-segmentation_amd", "dass_hip")
        for tag in sys.argv[1:] or ["hip", "exp1", "exp2", "exp3"]:
            env = dict(os.environ, X3_CHILD="1", DASS_HIP_LIB=os.path.join(d, "libdass_%s.so" % tag))
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            for l in r.stdout.splitlines():
                if l.startswith("STAMP "):
                    print(l, flush=True)
            res = [l for l in r.stdout.splitlines() if l.startswith("RES")]
            print("%-5s %s" % (tag, res[0][4:] if res else "FAILED " + r.stderr[-400:]), flush=True)
