#!/usr/bin/env python
"""How long the HOST needs to enqueue one train step (config A) versus how long the GPU needs to run it: the step is GPU-bound
only while the first is smaller.  Method: the GPU is made to lag behind (a long sleep kernel is not available, so a burst of
large matmuls is queued first), then K steps are enqueued without any synchronisation and the host time is taken.
    python tools/host_enqueue.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from dass_hip import ops  # noqa: E402
from dass_hip.optim import SGD  # noqa: E402
from models.deeplab import DeepLab  # noqa: E402
from utils.loss import SegmentationLosses  # noqa: E402
from oracle.deeplab_cpu import synthetic_batch  # noqa: E402


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    torch.manual_seed(1)
    model = DeepLab(backbone="resnet101", output_stride=16, num_classes=19, sync_bn=False, freeze_bn=False, pretrained=False).cuda().train()
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    opt = SGD([{"params": model.get_1x_lr_params(), "lr": 0.01}, {"params": model.get_10x_lr_params(), "lr": 0.1}], momentum=0.9, weight_decay=5e-4)
    x, y = synthetic_batch(8, 513, 513, 19, first_index=0)
    x, y = x.cuda(), y.cuda()

    def step():
        opt.zero_grad(set_to_none=True)
        crit(model(x), y).backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        step()
    torch.cuda.synchronize()
    gpu_bound = (time.perf_counter() - t0) / k
    a = torch.randn(8192, 8192, device="cuda")
    torch.cuda.synchronize()
    for _ in range(60):           # ~0.3 s of queued GPU work: the host runs ahead of the device from here on
        a = a @ a * 1e-4
    t0 = time.perf_counter()
    for _ in range(k):
        step()
    host = (time.perf_counter() - t0) / k
    torch.cuda.synchronize()
    print("train step: %.1f ms wall (synchronised run of %d steps); host enqueue alone %.1f ms per step" % (gpu_bound * 1e3, k, host * 1e3))


if __name__ == "__main__":
    main()
