#!/usr/bin/env python
"""Launch ONE forward conv shape on the pre-split engine a few times (for rocprofv3 --pmc passes; the conversion pass runs
once, outside the repeated launch).  usage: x3_one.py N H W C K ks stride pad dil [reps] [tile code]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
import torch  # noqa: E402
from dass_hip import ops  # noqa: E402
from dass_hip._lib import lib  # noqa: E402

n, h, w, c, k, ks, st, pad, dil = [int(v) for v in sys.argv[1:10]]
reps = int(sys.argv[10]) if len(sys.argv) > 10 else 5
if len(sys.argv) > 11:
    lib.dass_x3_force_tile(int(sys.argv[11]))
ops.set_f32_mma(os.environ.get("DASS_F32_MMA", "f16x3"))
oh, ow = ops.conv_out_size(h, ks, st, pad, dil), ops.conv_out_size(w, ks, st, pad, dil)
x = torch.randn((n, h, w, c), device="cuda")
wt = torch.randn((k, ks, ks, c), device="cuda") * 0.05
y = torch.empty((n, oh, ow, k), device="cuda")
x3, wop = ops.split3_rows(x, c, n * h * w, c), ops.prepare_conv_weight(wt, x3=True)
for _ in range(reps):
    ops.conv_x3_launch(x3, wop, y, k, (n, h, w, c, oh, ow, k, ks, ks, st, pad, dil))
torch.cuda.synchronize()
print("done", oh, ow)
