import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd")); sys.path.insert(0, ROOT)
import torch, torch.nn as nn
from dass_hip import ops
from models.backbone.mobilenet import InvertedResidual
from oracle import deeplab_cpu as O
torch.manual_seed(0)
for (cin, cout, stride, dil, t, hw) in [(160, 160, 1, 1, 6, 5), (96, 160, 2, 1, 6, 9), (160, 320, 1, 2, 6, 5), (32, 16, 1, 1, 1, 33), (24, 24, 1, 1, 6, 17)]:
    ob = O.OInvertedResidual(cin, cout, stride, dil, t)
    O.fill_state_dict(ob, seed=3)
    pb = InvertedResidual(cin, cout, stride, dil, t, nn.BatchNorm2d)
    pb.load_state_dict(ob.state_dict())
    for m in pb.modules():
        if isinstance(m, nn.Conv2d):
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    pb = pb.cuda()
    for train in (False, True):
        ob.train(train); pb.train(train)
        o64 = O.OInvertedResidual(cin, cout, stride, dil, t); o64.load_state_dict(ob.state_dict()); o64 = o64.double(); o64.train(train)
        x = torch.randn(2, cin, hw, hw)
        res = {}
        for tag, mod, xx in (("f64", o64, x.double()), ("f32", ob, x.clone()), ("hip", pb, x.cuda().contiguous(memory_format=torch.channels_last))):
            mod.zero_grad()
            xx = xx.requires_grad_(True)
            y = mod(xx)
            g = torch.Generator().manual_seed(5)
            go = torch.randn(y.shape, generator=g).to(y.dtype).to(y.device)
            y.backward(go)
            res[tag] = {"y": y.detach().double().cpu(), "dx": xx.grad.double().cpu(), **{k: p.grad.double().cpu() for k, p in mod.named_parameters()}}
        print("block", (cin, cout, stride, dil, t, hw), "train" if train else "eval")
        for k in res["f64"]:
            ref = res["f64"][k]
            e32 = (res["f32"][k] - ref).norm().item() / max(ref.norm().item(), 1e-12)
            ehip = (res["hip"][k] - ref).norm().item() / max(ref.norm().item(), 1e-12)
            flag = "  <<<<" if ehip > 10 * e32 + 1e-5 else ""
            print("   %-16s f32 %.2e  hip %.2e%s" % (k, e32, ehip, flag))
