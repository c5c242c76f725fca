"""end-to-end logits of both parity engines against an f64 run of the oracle, several weight seeds and both backbones:
max |dlogit|, argmax flips, flips outside near-ties (margin > 1e-3) -- evidence beyond the committed goldens"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd")); sys.path.insert(0, ROOT)
import torch
from dass_hip import ops
from models.deeplab import DeepLab
from oracle import deeplab_cpu as O

torch.set_num_threads(16)
for backbone, hw in (("resnet101", 193), ("mobilenet", 257)):
    for seed in (31, 32, 33):
        o64 = O.ODeepLab(backbone, 16, 19); O.fill_state_dict(o64, seed=seed); sd = {k: v.clone() for k, v in o64.state_dict().items()}
        o64 = o64.double().eval()
        x, _ = O.synthetic_batch(1, hw, hw, 19, first_index=700 + seed)
        with torch.no_grad():
            ref = o64(x.double())
            o32 = O.ODeepLab(backbone, 16, 19); o32.load_state_dict(sd); o32.eval()
            y32 = o32(x)
        top = ref.topk(2, dim=1)[0]; safe = (top[:, 0] - top[:, 1]) > 1e-3
        line = "%-9s %d^2 seed %d |logit|max %.1f  stock f32: err %.1e flips %d" % (backbone, hw, seed, ref.abs().max().item(), (y32.double() - ref).abs().max().item(),
                                                                                   int((y32.argmax(1) != ref.argmax(1)).sum()))
        for engine in ("bf16x6", "f32"):
            ops.set_f32_mma(engine)
            pm = DeepLab(backbone=backbone, num_classes=19, sync_bn=False, pretrained=False); pm.load_state_dict(sd); pm = pm.cuda().eval()
            with torch.no_grad():
                y = pm(x.cuda()).double().cpu()
            fl = (y.argmax(1) != ref.argmax(1))
            line += " | %s: err %.1e flips %d (outside near-ties %d)" % (engine, (y - ref).abs().max().item(), int(fl.sum()), int((fl & safe).sum()))
        print(line, flush=True)
