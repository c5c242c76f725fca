"""frozen-BN ResNet gradients vs the f64 oracle in each MFMA mode, several weight seeds: is a large worst-parameter
error rounding luck (one ReLU kink flip) or systematic?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from dass_hip import ops
from models.deeplab import DeepLab
from utils.loss import SegmentationLosses
from oracle import deeplab_cpu as O, selection_cpu as S

backbone, ncls, n, hw = "resnet", 19, 2, 65
for seed in (21, 22, 23):
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=500)
    m1, m2 = O.dropout_masks(n, 1, seed=22)
    o64 = O.ODeepLab(backbone, 16, ncls); O.fill_state_dict(o64, seed=seed); sd = {k: v.clone() for k, v in o64.state_dict().items()}
    o64 = o64.double().train()
    for m in o64.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    S.ce_loss(o64(x.double(), (m1[0].double(), m2[0].double())), lab).backward()
    g64 = {k: p.grad for k, p in o64.named_parameters()}
    floor = 1e-3 * float(np.median([v.norm().item() for v in g64.values()]))
    for mode in ("f32", "bf16x6", "bf16x3"):
        ops.set_f32_mma(mode)
        pm = DeepLab(backbone=backbone, num_classes=ncls, sync_bn=False, pretrained=False); pm.load_state_dict(sd); pm = pm.cuda().train(); pm.freeze_bn()
        crit = SegmentationLosses(cuda=True).build_loss("ce")
        loss = crit(pm(x.cuda(), dropout_masks=(m1[0].cuda(), m2[0].cuda())), lab.cuda()); loss.backward()
        rel = sorted((((p.grad.double().cpu() - g64[k]).norm().item() / max(g64[k].norm().item(), floor), k) for k, p in pm.named_parameters()), reverse=True)
        print("seed %d %-7s median %.2e worst: %s" % (seed, mode, float(np.median([r for r, _ in rel])), ", ".join("%s %.1e" % (k, r) for r, k in rel[:3])), flush=True)
