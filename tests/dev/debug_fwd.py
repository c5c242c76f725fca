import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd")); sys.path.insert(0, ROOT)
import torch, torch.nn as nn
from dass_hip import ops
from models.deeplab import DeepLab
from oracle import deeplab_cpu as O
om = O.ODeepLab("mobilenet", 16, 19); O.fill_state_dict(om, seed=21); om.eval()
o64 = O.ODeepLab("mobilenet", 16, 19); O.fill_state_dict(o64, seed=21); o64 = o64.double().eval()
pm = DeepLab(backbone="mobilenet", num_classes=19, sync_bn=False, pretrained=False); pm.load_state_dict(om.state_dict()); pm = pm.cuda().eval()
x, lab = O.synthetic_batch(2, 65, 65, 19, first_index=500)
acts = {"f64": {}, "f32": {}, "hip": {}}
def hook(tag, name):
    def f(mod, inp, out):
        o = out[0] if isinstance(out, tuple) else out
        acts[tag][name] = o.detach().double().cpu()
    return f
for tag, m in (("f64", o64), ("f32", om), ("hip", pm)):
    for i, blk in enumerate(m.backbone.features):
        blk.register_forward_hook(hook(tag, "features.%d" % i))
    m.aspp.register_forward_hook(hook(tag, "aspp"))
    m.decoder.register_forward_hook(hook(tag, "decoder"))
with torch.no_grad():
    y64 = o64(x.double()); y32 = om(x); yh = pm(x.cuda())
for name in acts["f64"]:
    if name not in acts["hip"]:
        continue
    ref = acts["f64"][name]
    e32 = (acts["f32"][name] - ref).norm().item() / ref.norm().item()
    eh = (acts["hip"][name] - ref).norm().item() / ref.norm().item()
    sat = float((ref >= 6).double().mean()) if "features" in name else 0.0
    print("%-14s f32 %.2e hip %.2e  max|ref| %.2f  frac>=6 %.3f" % (name, e32, eh, ref.abs().max().item(), sat))
print("logits f32 %.2e hip %.2e" % ((y32.double() - y64).abs().max().item(), (yh.double().cpu() - y64).abs().max().item()))
