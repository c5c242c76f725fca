"""How many ReLU gates of a DeepLab forward differ from the f64 oracle's, per conv engine and seed?  A gate flips when a
pre-activation lies within the engine's rounding error of zero; every flip moves the gradients of all layers upstream of it by
~1e-3.  Diagnostic for the true-ReLU gradient tests: python tools/gate_flips.py [backbone] [seeds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dass_hip import ops  # noqa: E402
from models.deeplab import DeepLab  # noqa: E402
from oracle import deeplab_cpu as O  # noqa: E402

backbone = sys.argv[1] if len(sys.argv) > 1 else "resnet"
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
train_bn = len(sys.argv) > 3 and sys.argv[3] == "train"
ncls, n, hw = 19, 2, 65
orig_cba, orig_relu = ops.conv_bn_act, torch.nn.functional.relu
for seed in range(31, 31 + seeds):
    om = O.ODeepLab(backbone, 16, ncls)
    O.fill_state_dict(om, seed=seed, randomize_bn_stats=not train_bn)
    o64 = O.ODeepLab(backbone, 16, ncls)
    o64.load_state_dict(om.state_dict())
    o64 = o64.double().train()
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=500 + seed)
    m1, m2 = O.dropout_masks(n, 1, seed=seed)
    if not train_bn:
        for m in o64.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eval()
    ref = []

    def relu(t, inplace=False):
        out = orig_relu(t)
        ref.append((out > 0))
        return out

    torch.nn.functional.relu = relu
    with torch.no_grad():
        o64(x.double(), (m1[0].double(), m2[0].double()))
    torch.nn.functional.relu = orig_relu
    line = "seed %d (%d ReLU sites, %d units):" % (seed, len(ref), sum(g.numel() for g in ref))
    for engine in ("bf16x6", "f16x3", "f32"):
        ops.set_f32_mma(engine)
        pm = DeepLab(backbone=backbone, output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False)
        pm.load_state_dict(om.state_dict())
        pm = pm.cuda().train()
        if not train_bn:
            pm.freeze_bn()
        got = []

        def wrapped(xx, conv, bn=None, act=ops.ACT_NONE, **kw):
            out = orig_cba(xx, conv, bn, act, **kw)
            if act == ops.ACT_RELU:
                first = out[0] if isinstance(out, tuple) else out
                got.append((first.detach() > 0).cpu())
            return out

        ops.conv_bn_act = wrapped
        with torch.no_grad():
            pm(x.cuda(), dropout_masks=(m1[0].cuda(), m2[0].cuda()))
        ops.conv_bn_act = orig_cba
        # match by shape in call order (the oracle's extra ReLU sites -- ASPP pool branch -- are skipped)
        used, flips, where = [False] * len(ref), 0, []
        for gi, g in enumerate(got):
            for i, r in enumerate(ref):
                if not used[i] and tuple(r.shape) == tuple(g.shape):
                    used[i] = True
                    d = int((r != g).sum())
                    flips += d
                    if d:
                        where.append("%d:%d" % (gi, d))
                    break
        line += "  %s %d flips [%s]" % (engine, flips, " ".join(where[:8]))
    print(line, flush=True)
