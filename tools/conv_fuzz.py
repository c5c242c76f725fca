#!/usr/bin/env python
"""random conv shapes (channels, sizes, kernel, stride, padding, dilation, bias / BN / residual / activation) through
ops.conv_bn_act in every engine, forward and all gradients against an f64 torch reference:  conv_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "deep-active-semantic-segmentation_amd"))
import random
import torch
import torch.nn as nn
import torch.nn.functional as F
from dass_hip import ops

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {}
for i in range(cases):
    n = rng.choice([1, 2, 3]); c = rng.choice([4, 8, 16, 24, 36, 64, 96, 160, 304]); k = rng.choice([4, 8, 20, 32, 48, 64, 100, 128, 256])
    ks = rng.choice([1, 1, 3, 3, 3, 5]); stride = rng.choice([1, 1, 1, 2]); dil = rng.choice([1, 1, 2, 3, 6]) if ks > 1 else 1
    pad = rng.choice([0, dil * (ks // 2), dil * (ks // 2), rng.randint(0, 3)])
    h, w = rng.randint(5, 40), rng.randint(5, 40)
    if (h + 2 * pad - dil * (ks - 1) - 1) // stride + 1 <= 0 or (w + 2 * pad - dil * (ks - 1) - 1) // stride + 1 <= 0:
        continue
    use_bn, train, use_res, act = rng.random() < 0.6, rng.random() < 0.5, rng.random() < 0.3, rng.choice([ops.ACT_NONE, ops.ACT_RELU, ops.ACT_RELU6])
    bias = (not use_bn) and rng.random() < 0.5
    g = torch.Generator().manual_seed(1000 + i)
    x = torch.randn(n, c, h, w, generator=g)
    conv = nn.Conv2d(c, k, ks, stride, pad, dil, bias=bias)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (c * ks * ks)) ** 0.5)
    bn = nn.BatchNorm2d(k) if use_bn else None
    if bn is not None:
        with torch.no_grad():
            bn.weight.copy_(torch.rand(k, generator=g) + 0.5); bn.bias.copy_(torch.randn(k, generator=g) * 0.2)
            bn.running_mean.copy_(torch.randn(k, generator=g) * 0.1); bn.running_var.copy_(torch.rand(k, generator=g) + 0.5)
        bn.train(train)
    # f64 reference
    c64, b64 = conv.double(), (bn.double() if bn is not None else None)
    import copy
    c64 = copy.deepcopy(conv).double(); b64 = copy.deepcopy(bn).double() if bn is not None else None
    xr = x.double().requires_grad_(True)
    y = c64(xr)
    res = torch.randn(y.shape, generator=g) if use_res else None
    rr = res.double().requires_grad_(True) if use_res else None
    if b64 is not None:
        y = b64(y)
    if rr is not None:
        y = y + rr
    y = F.relu(y) if act == ops.ACT_RELU else (F.relu6(y) if act == ops.ACT_RELU6 else y)
    go = torch.randn(y.shape, generator=g)
    y.backward(go.double())
    for engine in ("bf16x6", "f32"):
        ops.set_f32_mma(engine)
        cd = copy.deepcopy(conv).float().cuda(); cd.weight.data = cd.weight.data.contiguous(memory_format=torch.channels_last)
        bd = copy.deepcopy(bn).float().cuda() if bn is not None else None
        if bd is not None:
            bd.train(train)
        xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        rd = res.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) if use_res else None
        yd = ops.conv_bn_act(xd, cd, bd, act, residual=rd)
        yd.backward(go.cuda().contiguous(memory_format=torch.channels_last))
        torch.cuda.synchronize()
        def rel(a, b):
            return ((a.detach().double().cpu() - b).norm() / max(b.norm().item(), 1e-12)).item()
        errs = {"y": rel(yd, y.detach()), "dx": rel(xd.grad, xr.grad), "dw": rel(cd.weight.grad, c64.weight.grad)}
        if use_res:
            errs["dres"] = rel(rd.grad, rr.grad)
        if bd is not None:
            errs["dgamma"] = rel(bd.weight.grad, b64.weight.grad)
        if bias:
            errs["dbias"] = rel(cd.bias.grad, c64.bias.grad)
        # relu kinks: a value within rounding of 0 / 6 may flip its gate; forward must always be tight
        bad = errs["y"] > 2e-5 or max(errs.values()) > 5e-3
        for kk, v in errs.items():
            worst[(engine, kk)] = max(worst.get((engine, kk), 0.0), v)
        if bad:
            print("CASE %d %s n%d c%d k%d ks%d s%d p%d d%d %dx%d bn=%s train=%s res=%s act=%d bias=%s -> %s"
                  % (i, engine, n, c, k, ks, stride, pad, dil, h, w, use_bn, train, use_res, act, bias, {a: "%.1e" % b for a, b in errs.items()}), flush=True)
print("cases", cases, "worst rel-L2 per engine/tensor:", {("%s/%s" % k): "%.1e" % v for k, v in sorted(worst.items())})
