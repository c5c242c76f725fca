#!/usr/bin/env python
"""300 GRAPHED training steps of DeepLab-R101 513^2 batch 8 on a learnable target with the reference's poly learning-rate schedule running
through the replays (active_train.py:101), next to the same 300 steps eager: loss trend, final loss, allocator footprint, images/s.
python tools/soak_graph.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
import torch  # noqa: E402
from dass_hip.graph import GraphedStep  # noqa: E402
from dass_hip.optim import SGD  # noqa: E402
from models.deeplab import DeepLab  # noqa: E402
from utils.loss import SegmentationLosses  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
y = (torch.arange(513, device="cuda") * 19 // 513).float().view(1, 1, 513).expand(8, 513, 513).contiguous()
torch.manual_seed(1)
x = torch.randn(8, 3, 513, 513, device="cuda") * 0.3 + (y / 9.0 - 1.0).unsqueeze(1)
for mode in ("graph", "eager"):
    torch.manual_seed(0)
    m = DeepLab(backbone="resnet101", output_stride=16, num_classes=19, sync_bn=False, freeze_bn=False, pretrained=False).cuda().train()
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    opt = SGD([{"params": m.get_1x_lr_params(), "lr": 0.01}, {"params": m.get_10x_lr_params(), "lr": 0.1}], momentum=0.9, weight_decay=5e-4)

    def set_lr(i):
        f = (1.0 - i / float(steps)) ** 0.9
        opt.param_groups[0]["lr"], opt.param_groups[1]["lr"] = 0.01 * f, 0.1 * f

    def step():
        opt.zero_grad(set_to_none=True)
        loss = crit(m(x), y)
        loss.backward()
        opt.step()
        return loss

    run = step
    done = 0
    if mode == "graph":
        for i in range(2):
            set_lr(i)
            step()
        done = 2
        run = GraphedStep(step, warmup=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(done, steps):
        set_lr(i)
        loss = run()
        if i % 50 == 49 or i == steps - 1:
            torch.cuda.synchronize()
            print("%s step %3d loss %.5f lr %.5f allocated %.2f GB reserved %.2f GB" % (mode, i + 1, float(loss), opt.param_groups[0]["lr"],
                  torch.cuda.memory_allocated() / 2 ** 30, torch.cuda.memory_reserved() / 2 ** 30), flush=True)
    torch.cuda.synchronize()
    print("%s: %.1f images/s over %d steps" % (mode, 8 * (steps - done) / (time.perf_counter() - t0), steps - done), flush=True)
    if mode == "graph":
        run.release()
    del m, opt, run
    torch.cuda.empty_cache()
