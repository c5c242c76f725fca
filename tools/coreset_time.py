#!/usr/bin/env python
"""BASELINE config E on one GPU: core-set selection (encoder features of the pool -> 2736-d vectors -> k-center greedy)
on a synthetic Cityscapes-shaped pool resident in HBM:  coreset_time.py [pool_images] [picks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
import torch
from dass_hip import ops
from dass_hip.dist import ModuleWrapper
from models.deeplab import DeepLab
from active_selection.core_set import ActiveSelectionCoreSet

npool = int(sys.argv[1]) if len(sys.argv) > 1 else 96
picks = int(sys.argv[2]) if len(sys.argv) > 2 else 12
torch.manual_seed(0)
model = ModuleWrapper(DeepLab(backbone="resnet101", output_stride=16, num_classes=19, sync_bn=False, pretrained=False).cuda().eval())
keys = [("pool_%06d" % i).encode() for i in range(npool)]
pool = {}
for i, k in enumerate(keys):
    g = torch.Generator().manual_seed(1000 + i)
    pool[k] = (torch.randn(1, 3, 513, 513, generator=g).cuda(), torch.zeros(1, 513, 513).cuda())
b = 8


def factory(images, include_labels=False):
    for i in range(0, len(images), b):
        chunk = images[i:i + b]
        yield {"image": torch.cat([pool[k][0] for k in chunk]), "label": torch.cat([pool[k][1] for k in chunk])}


sel = ActiveSelectionCoreSet(None, 513, b, loader_factory=factory)
lab, unl = keys[: npool // 8], keys[npool // 8:]
sel.get_k_center_greedy_selections(2, model, unl[:16], lab[:4])  # warm-up
torch.cuda.synchronize(); t0 = time.perf_counter()
out = sel.get_k_center_greedy_selections(picks, model, unl, lab)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("core-set: %d pool images (%d labeled), %d picks in %.3f s = %.1f pool images/s (engine %s)" % (npool, len(lab), len(out), dt, npool / dt, ops.f32_mma()))
