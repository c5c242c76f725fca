#!/usr/bin/env python
"""60 training steps of DeepLab-R101 513^2 batch 8: loss trend and allocator footprint (the weight-gradient arenas and the
split-weight operands must not grow): prints every 10 steps"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
import torch
from dass_hip import ops
from models.deeplab import DeepLab
from utils.loss import SegmentationLosses
torch.manual_seed(0)
m = DeepLab(backbone="resnet101", output_stride=16, num_classes=19, sync_bn=False, freeze_bn=False, pretrained=False).cuda().train()
crit = SegmentationLosses(cuda=True).build_loss("ce")
opt = torch.optim.SGD([{"params": m.get_1x_lr_params(), "lr": 0.01}, {"params": m.get_10x_lr_params(), "lr": 0.1}], momentum=0.9, weight_decay=5e-4)
x = torch.randn(8, 3, 513, 513, device="cuda")
if "--learnable" in sys.argv:   # a target the net can fit: 19 vertical bands, brightened in the image
    y = (torch.arange(513, device="cuda") * 19 // 513).float().view(1, 1, 513).expand(8, 513, 513).contiguous()
    x = x * 0.3 + (y / 9.0 - 1.0).unsqueeze(1)
else:
    y = torch.randint(0, 19, (8, 513, 513), device="cuda").float()
for i in range(60):
    opt.zero_grad(set_to_none=True)
    loss = crit(m(x), y); loss.backward(); opt.step()
    if i % 10 == 9:
        torch.cuda.synchronize()
        print("step %d loss %.4f allocated %.2f GB reserved %.2f GB" % (i + 1, loss.item(), torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30), flush=True)
