#!/usr/bin/env python
"""Launch ONE conv shape a few times (for rocprofv3 --pmc passes).  usage: conv_one.py N H W C K ks stride pad dil [reps] [mode]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
import torch  # noqa: E402
from dass_hip import ops  # noqa: E402
from dass_hip._lib import check, lib  # noqa: E402

n, h, w, c, k, ks, st, pad, dil = [int(v) for v in sys.argv[1:10]]
reps = int(sys.argv[10]) if len(sys.argv) > 10 else 5
mode = sys.argv[11] if len(sys.argv) > 11 else "fwd"
oh, ow = ops.conv_out_size(h, ks, st, pad, dil), ops.conv_out_size(w, ks, st, pad, dil)
x = torch.randn((n, h, w, c), device="cuda")
wt = torch.randn((k, ks, ks, c), device="cuda") * 0.05
y = torch.empty((n, oh, ow, k), device="cuda")
dy = torch.randn((n, oh, ow, k), device="cuda")
dw = torch.empty((k, ks, ks, c), device="cuda")
wop = ops.prepare_conv_weight(wt)
for _ in range(reps):
    if mode == "fwd":
        ops.conv_launch(x, c, wop, y, k, (n, h, w, c, oh, ow, k, ks, ks, st, pad, dil))
    else:
        check(lib.dass_conv2d_wgrad(ops._p(x), c, ops._p(dy), k, ops._p(dw), n, h, w, c, oh, ow, k, ks, ks, st, pad, dil, ops._cdt(dy), ops._stream()), "wgrad")
torch.cuda.synchronize()
print("done", oh, ow)
