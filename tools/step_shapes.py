#!/usr/bin/env python
"""Per-layer-shape kernel times of the conv entry points INSIDE a real train step (R101 513^2 batch 8, f16x3): the library's launch
profile (dass_hip/_lib.py:KernelTimer) keyed by (entry point, M, C, K, taps) -- next to tools/x3_time.py's isolated numbers this
shows what a launch loses to its neighbours in the step.   python tools/step_shapes.py [batch] [size]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import synthetic_batch  # noqa: E402
from dass_hip import ops  # noqa: E402
from dass_hip._lib import BN_ENTRY_POINTS, KernelTimer  # noqa: E402
from dass_hip.optim import SGD  # noqa: E402
from models.deeplab import DeepLab  # noqa: E402
from utils.loss import SegmentationLosses  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 8
s = int(sys.argv[2]) if len(sys.argv) > 2 else 513
ops.set_compute_dtype(torch.float32)
ops.set_f32_mma("f16x3")
torch.manual_seed(1234)
model = DeepLab(backbone="resnet101", output_stride=16, num_classes=19, sync_bn=False, pretrained=False).cuda().train()
crit = SegmentationLosses(cuda=True).build_loss("ce")
opt = SGD([{"params": model.get_1x_lr_params(), "lr": 0.01}, {"params": model.get_10x_lr_params(), "lr": 0.1}], momentum=0.9, weight_decay=5e-4)
x, y = synthetic_batch(b, s, s, 19, 0)
x, y = x.cuda(), y.cuda()


def step():
    opt.zero_grad(set_to_none=True)
    crit(model(x), y).backward()
    opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
reps = 3
with KernelTimer() as kt:
    step()
    torch.cuda.synchronize()
    kt.restart()
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    kernels, calls = kt.results()
    shapes = list(kt.shapes)
by = {}
for (name, tag, work, ms, knames), shp in zip(calls, shapes):
    if name in BN_ENTRY_POINTS or shp is None:
        continue
    n, oh, ow, c, k, r = shp
    e = by.setdefault((name.replace("dass_conv2d_", ""), n * oh * ow, c, k, r, tag), [0, 0.0, 0.0])
    e[0] += 1
    e[1] += ms
    e[2] += work
print("%-22s %8s %5s %5s %2s %-10s | %5s %8s %8s %7s" % ("entry point", "M", "C", "K", "R", "tile", "n/step", "avg us", "ms/step", "TF/s"))
for key, v in sorted(by.items(), key=lambda kv: -kv[1][1]):
    name, m, c, k, r, tag = key
    tile = "%dx%d%s" % (tag >> 16, (tag >> 4) & 0xfff, "w" if tag & 2 else ("s" if tag & 1 else "")) if tag else "-"
    print("%-22s %8d %5d %5d %2d %-10s | %5.1f %8.1f %8.3f %7.1f" % (name, m, c, k, r, tile, v[0] / reps, 1e3 * v[1] / v[0], v[1] / reps, v[2] / v[1]))
