#!/usr/bin/env python
"""Where the HOST's time goes in one eager train step (R101 513^2 batch 8): cProfile over 5 steps, top functions by own time.
The step launches ~550 kernels; its host cost (26-29 ms, box dependent) is at the level of its GPU time.   python tools/host_profile.py"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import synthetic_batch  # noqa: E402
from dass_hip import ops  # noqa: E402
from dass_hip.optim import SGD  # noqa: E402
from models.deeplab import DeepLab  # noqa: E402
from utils.loss import SegmentationLosses  # noqa: E402

ops.set_compute_dtype(torch.float32)
ops.set_f32_mma(os.environ.get("DASS_F32_MMA", "f16x3"))
torch.manual_seed(1234)
model = DeepLab(backbone="resnet101", output_stride=16, num_classes=19, sync_bn=False, pretrained=False).cuda().train()
crit = SegmentationLosses(cuda=True).build_loss("ce")
opt = SGD([{"params": model.get_1x_lr_params(), "lr": 0.01}, {"params": model.get_10x_lr_params(), "lr": 0.1}], momentum=0.9, weight_decay=5e-4)
x, y = synthetic_batch(8, 513, 513, 19, 0)
x, y = x.cuda(), y.cuda()


def step():
    opt.zero_grad(set_to_none=True)
    crit(model(x), y).backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
# host-only time of a step: issue 5 steps without waiting for the GPU between them, on a small input the GPU finishes early
t0 = time.perf_counter()
for _ in range(5):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("5 steps issued in %.1f ms (%.1f ms per step of host time, GPU not waited for)" % ((t1 - t0) * 1e3, (t1 - t0) * 200))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)

# the backward pass runs on autograd's device thread, which the profiler above does not see: profile it from inside, by
# wrapping every autograd.Function's backward (looked up on the class at call time)
prb = cProfile.Profile()


def wrap(cls):
    orig = cls.__dict__["backward"].__func__

    def backward(*a):
        prb.enable()
        try:
            return orig(*a)
        finally:
            prb.disable()
    cls.backward = staticmethod(backward)


def all_subclasses(c):
    for s in c.__subclasses__():
        yield s
        yield from all_subclasses(s)


for cls in set(all_subclasses(torch.autograd.Function)):
    if "backward" in cls.__dict__ and isinstance(cls.__dict__["backward"], staticmethod) and cls.__module__.split(".")[0] in ("dass_hip", "models", "utils"):
        wrap(cls)
for _ in range(5):
    step()
torch.cuda.synchronize()
print("---- backward functions (autograd thread), 5 steps ----")
pstats.Stats(prb).sort_stats("tottime").print_stats(40)
