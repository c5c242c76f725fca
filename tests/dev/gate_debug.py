"""debug: per-parameter gradient error (network order) of the gate-injected comparison in train-mode BN"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_grad_parity_gpu import _GateReplay, _setup
ops, O, S = _setup()
from models.deeplab import DeepLab
from utils.loss import SegmentationLosses
ops.set_x3_pipeline("off")
ncls, n, hw = 19, 2, 65
om = O.ODeepLab("resnet", 16, ncls); O.fill_state_dict(om, seed=21)
pm = DeepLab(backbone="resnet", output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False); pm.load_state_dict(om.state_dict()); pm = pm.cuda().train()
o64 = O.ODeepLab("resnet", 16, ncls); o64.load_state_dict(om.state_dict()); o64 = o64.double().train()
x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=500)
m1, m2 = O.dropout_masks(n, 1, seed=22)
rec = _GateReplay(ops)
rec.record()
loss = SegmentationLosses(cuda=True).build_loss("ce")(pm(x.cuda(), dropout_masks=(m1[0].cuda(), m2[0].cuda())), lab.cuda()); loss.backward()
rec.stop_recording(); rec.replay()
lo = S.ce_loss(o64(x.double(), (m1[0].double(), m2[0].double())), lab); lo.backward()
rec.restore()
g64 = {k: p.grad for k, p in o64.named_parameters()}
for k, p in reversed(list(pm.named_parameters())):
    g = p.grad.double().cpu(); r = g64[k]
    cos = float((g * r).sum() / (g.norm() * r.norm() + 1e-300))
    print("%-48s rel %.2e  ratio-1 %+.2e  1-cos %.1e  |g| %.2e" % (k, (g - r).norm().item() / max(r.norm().item(), 1e-30), g.norm().item() / max(r.norm().item(), 1e-30) - 1, 1 - cos, r.norm().item()))
