"""Op-level parity of the HIP kernels (through the C-ABI) against plain PyTorch fp32 on the CPU.

Each case runs the torch op the reference calls at that site (F.conv2d, F.batch_norm, F.relu,
F.max_pool2d, F.interpolate, F.cross_entropy ...) on the CPU and the libdass_hip path on cuda:0 with the
same seeded inputs.  Tolerances: forward 2e-4 relative to the tensor's max magnitude (f32 MFMA is an
exact f32 fma chain; only summation order differs), gradients 5e-4.
"""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ops():
    from dass_hip import ops

    ops.set_compute_dtype(torch.float32)
    return ops


def _close(a, b, tol, what=""):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(b.abs().max().item(), 1e-6)
    err = (a - b).abs().max().item() / scale
    assert err <= tol, "%s: rel err %.3e > %.1e (max |ref| %.3e)" % (what, err, tol, scale)


def _cl(t):
    return t.cuda().contiguous(memory_format=torch.channels_last)


CONV_CASES = [
    # (N, C, H, W, K, ksize, stride, pad, dil)
    (2, 64, 17, 17, 64, 1, 1, 0, 1),
    (2, 32, 19, 23, 96, 3, 1, 1, 1),
    (1, 128, 33, 33, 256, 3, 1, 6, 6),
    (2, 64, 33, 33, 64, 3, 1, 12, 12),
    (1, 64, 33, 33, 32, 3, 1, 18, 18),
    (2, 48, 21, 21, 128, 3, 2, 1, 1),
    (2, 256, 9, 9, 512, 1, 2, 0, 1),
    (2, 24, 13, 13, 144, 1, 1, 0, 1),
    (1, 304, 17, 17, 256, 3, 1, 1, 1),
    (2, 512, 9, 9, 512, 3, 1, 2, 2),
    (3, 64, 9, 9, 48, 1, 1, 2, 1),  # 1x1 over a zero-padded input (MobileNet fixed_padding before expand)
    (1, 2048, 5, 5, 256, 1, 1, 0, 1),
]


@pytest.fixture(autouse=True)
def _restore_mma_mode():
    from dass_hip import ops

    mode = ops.f32_mma()
    yield
    ops.set_f32_mma(mode)


@pytest.mark.parametrize("mode", ["f32", "bf16x6", "bf16x3"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_backward(case, mode):
    ops = _ops()
    ops.set_f32_mma(mode)
    n, c, h, w, k, ks, stride, pad, dil = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(n, c, h, w, generator=g)
    conv = nn.Conv2d(c, k, ks, stride, pad, dil, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (c * ks * ks)) ** 0.5)
    xr = x.clone().requires_grad_(True)
    yr = conv(xr)
    go = torch.randn(yr.shape, generator=g)
    yr.backward(go)

    conv_d = nn.Conv2d(c, k, ks, stride, pad, dil, bias=False).cuda()
    with torch.no_grad():
        conv_d.weight.copy_(conv.weight)
    conv_d.weight.data = conv_d.weight.data.contiguous(memory_format=torch.channels_last)
    xd = _cl(x).requires_grad_(True)
    yd = ops.conv_bn_act(xd, conv_d)
    _close(yd, yr, 2e-4, "conv fwd %s" % (case,))
    yd.backward(_cl(go))
    _close(xd.grad, xr.grad, 5e-4, "conv dgrad %s" % (case,))
    _close(conv_d.weight.grad, conv.weight.grad, 5e-4, "conv wgrad %s" % (case,))


@pytest.mark.parametrize("case", [(2, 304, 33, 33, 256, 3, 1, 1, 1), (2, 256, 33, 33, 256, 3, 1, 2, 2), (4, 1024, 17, 17, 256, 1, 1, 0, 1),
                                  (2, 128, 33, 33, 128, 3, 2, 1, 1), (2, 64, 65, 65, 64, 3, 1, 1, 1), (1, 2048, 9, 9, 256, 3, 1, 12, 12),
                                  (2, 36, 21, 19, 40, 3, 1, 1, 1)])
def test_conv_mma_modes_vs_f64(case):
    """The three ways of multiplying f32 tensors (ops.set_f32_mma) against an f64 torch conv, relative L2 error of
    forward, input gradient and weight gradient.  'bf16x6' (three-way bf16 split, six products) must sit at the f32
    rounding level -- it is the default parity engine; 'bf16x3' (two-way split) carries 17-bit products (~4.5e-6)."""
    ops = _ops()
    n, c, h, w, k, ks, stride, pad, dil = case
    g = torch.Generator().manual_seed(1234 + c + k)
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(k, c, ks, ks, generator=g) * (2.0 / (c * ks * ks)) ** 0.5
    x64 = x.double().requires_grad_(True)
    w64 = wt.double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, stride, pad, dil)
    go = torch.randn(y64.shape, generator=g)
    y64.backward(go.double())
    rel = lambda a, b: ((a.detach().double().cpu() - b).norm() / b.norm()).item()  # noqa: E731
    errs = {}
    for mode in ("f32", "bf16x6", "bf16x3"):
        ops.set_f32_mma(mode)
        conv_d = nn.Conv2d(c, k, ks, stride, pad, dil, bias=False).cuda()
        with torch.no_grad():
            conv_d.weight.copy_(wt)
        conv_d.weight.data = conv_d.weight.data.contiguous(memory_format=torch.channels_last)
        xd = _cl(x).requires_grad_(True)
        yd = ops.conv_bn_act(xd, conv_d)
        yd.backward(_cl(go))
        errs[mode] = (rel(yd, y64.detach()), rel(xd.grad, x64.grad), rel(conv_d.weight.grad, w64.grad))
    print(case, {m: ["%.1e" % v for v in e] for m, e in errs.items()})
    for i, what in enumerate(("fwd", "dgrad", "wgrad")):
        assert errs["f32"][i] <= 3e-6, (what, errs)
        assert errs["bf16x6"][i] <= max(2.0 * errs["f32"][i], 1e-6), (what, errs)   # f32-exact products: summation order only
        assert errs["bf16x3"][i] <= 1.2e-5, (what, errs)


def test_conv_stem_image_input():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 65, 65, generator=g)
    conv = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
    bn = nn.BatchNorm2d(64)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(64, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(64, generator=g) * 0.1)
    ref = F.relu(bn(conv(x)))
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    import copy

    conv_d, bn_d = copy.deepcopy(conv).cuda(), copy.deepcopy(bn).cuda()
    conv_d.weight.grad = None
    bn_d.weight.grad = bn_d.bias.grad = None
    bn_d.running_mean.zero_()
    bn_d.running_var.fill_(1)
    bn_d.num_batches_tracked.zero_()
    out = ops.conv_bn_act(x.cuda(), conv_d, bn_d, ops.ACT_RELU, image_input=True)
    _close(out, ref, 2e-4, "stem fwd")
    out.backward(_cl(go))
    _close(conv_d.weight.grad, conv.weight.grad, 5e-4, "stem wgrad")
    _close(bn_d.weight.grad, bn.weight.grad, 5e-4, "stem dgamma")
    _close(bn_d.bias.grad, bn.bias.grad, 5e-4, "stem dbeta")
    _close(bn_d.running_mean, bn.running_mean, 1e-4, "running_mean")
    _close(bn_d.running_var, bn.running_var, 1e-4, "running_var")


@pytest.mark.parametrize("act", ["relu", "relu6"])
@pytest.mark.parametrize("train", [True, False])
def test_conv_bn_act_residual(act, train):
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    n, c, h, w, k = 2, 64, 15, 15, 128
    x = torch.randn(n, c, h, w, generator=g)
    res = torch.randn(n, k, h, w, generator=g)
    conv = nn.Conv2d(c, k, 3, 1, 2, 2, bias=False)
    bn = nn.BatchNorm2d(k)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(k, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(k, generator=g) * 0.2)
        bn.running_mean.copy_(torch.randn(k, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(k, generator=g) + 0.5)
    import copy

    conv_d, bn_d = copy.deepcopy(conv).cuda(), copy.deepcopy(bn).cuda()
    bn.train(train)
    bn_d.train(train)
    actf = F.relu if act == "relu" else F.relu6
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    ref = actf(bn(conv(xr)) + rr)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    xd, rd = _cl(x).requires_grad_(True), _cl(res).requires_grad_(True)
    out = ops.conv_bn_act(xd, conv_d, bn_d, ops.ACT_RELU if act == "relu" else ops.ACT_RELU6, residual=rd)
    _close(out, ref, 2e-4, "fwd")
    out.backward(_cl(go))
    _close(xd.grad, xr.grad, 5e-4, "dx")
    _close(rd.grad, rr.grad, 5e-4, "dres")
    _close(conv_d.weight.grad, conv.weight.grad, 5e-4, "dw")
    _close(bn_d.weight.grad, bn.weight.grad, 5e-4, "dgamma")
    _close(bn_d.bias.grad, bn.bias.grad, 5e-4, "dbeta")
    if train:
        _close(bn_d.running_var, bn.running_var, 1e-4, "running_var")
    # inference (no grad) takes the fully fused epilogue
    bn.eval()
    bn_d.eval()
    with torch.no_grad():
        ref2 = actf(bn(conv(x)) + res)
        out2 = ops.conv_bn_act(_cl(x), conv_d, bn_d, ops.ACT_RELU if act == "relu" else ops.ACT_RELU6, residual=_cl(res))
    _close(out2, ref2, 2e-4, "fused eval fwd")


def test_classifier_bias_19():
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 256, 11, 11, generator=g)
    conv = nn.Conv2d(256, 19, 1)
    xr = x.clone().requires_grad_(True)
    ref = conv(xr)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    import copy

    conv_d = copy.deepcopy(conv).cuda()
    conv_d.weight.grad = conv_d.bias.grad = None
    xd = _cl(x).requires_grad_(True)
    out = ops.conv_bn_act(xd, conv_d)
    _close(out, ref, 2e-4, "cls fwd")
    up_ref = F.interpolate(ref, size=(41, 41), mode="bilinear", align_corners=True)
    up = ops.upsample_to_nchw(out, 41, 41)
    _close(up, up_ref, 2e-4, "final upsample")
    out.backward(go.cuda())
    _close(xd.grad, xr.grad, 5e-4, "cls dx")
    _close(conv_d.weight.grad, conv.weight.grad, 5e-4, "cls dw")
    _close(conv_d.bias.grad, conv.bias.grad, 5e-4, "cls dbias")


@pytest.mark.parametrize("cfg", [(32, 1, 1), (96, 2, 1), (144, 1, 2), (576, 1, 4)])
def test_depthwise(cfg):
    ops = _ops()
    c, stride, dil = cfg
    g = torch.Generator().manual_seed(c)
    x = torch.randn(2, c, 17, 19, generator=g)
    conv = nn.Conv2d(c, c, 3, stride, 0, dil, groups=c, bias=False)
    bn = nn.BatchNorm2d(c)
    xr = x.clone().requires_grad_(True)
    ref = F.relu6(bn(conv(F.pad(xr, (dil, dil, dil, dil)))))
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    import copy

    conv_d, bn_d = copy.deepcopy(conv).cuda(), copy.deepcopy(bn).cuda()
    conv_d.weight.grad = None
    bn_d.weight.grad = bn_d.bias.grad = None
    bn_d.running_mean.zero_()
    bn_d.running_var.fill_(1)
    xd = _cl(x).requires_grad_(True)
    out = ops.conv_bn_act(xd, conv_d, bn_d, ops.ACT_RELU6, extra_pad=dil)
    _close(out, ref, 2e-4, "dw fwd")
    out.backward(_cl(go))
    _close(xd.grad, xr.grad, 5e-4, "dw dx")
    _close(conv_d.weight.grad, conv.weight.grad, 5e-4, "dw dw")
    _close(bn_d.weight.grad, bn.weight.grad, 5e-4, "dw dgamma")


def test_maxpool_and_add():
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 64, 33, 31, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = F.max_pool2d(xr, 3, 2, 1)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    xd = _cl(x).requires_grad_(True)
    out = ops.maxpool3x3s2(xd)
    _close(out, ref, 0, "maxpool fwd")
    out.backward(_cl(go))
    _close(xd.grad, xr.grad, 1e-6, "maxpool bwd")
    a, b = torch.randn(2, 32, 5, 5, generator=g), torch.randn(2, 32, 5, 5, generator=g)
    _close(ops.add(_cl(a), _cl(b)), a + b, 1e-6, "add")


def test_upsample_cat_and_concat():
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 256, 9, 9, generator=g)
    low = torch.randn(2, 48, 33, 33, generator=g)
    xr, lr = x.clone().requires_grad_(True), low.clone().requires_grad_(True)
    ref = torch.cat((F.interpolate(xr, size=(33, 33), mode="bilinear", align_corners=True), lr), 1)
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    xd, ld = _cl(x).requires_grad_(True), _cl(low).requires_grad_(True)
    out = ops.upsample_cat(xd, ld)
    _close(out, ref, 1e-5, "upsample_cat fwd")
    out.backward(_cl(go))
    _close(xd.grad, xr.grad, 1e-4, "upsample bwd")
    _close(ld.grad, lr.grad, 1e-6, "cat bwd")
    parts = [torch.randn(2, 256, 7, 7, generator=g) for _ in range(5)]
    pd = [_cl(p).requires_grad_(True) for p in parts]
    cat = ops.concat(*pd)
    _close(cat, torch.cat(parts, 1), 0, "concat")
    go2 = torch.randn(cat.shape, generator=g)
    cat.backward(_cl(go2))
    _close(pd[3].grad, go2[:, 768:1024], 0, "concat bwd")


@pytest.mark.parametrize("train", [True, False])
def test_aspp_pool_branch(train):
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(3, 320, 9, 9, generator=g)
    conv = nn.Conv2d(320, 256, 1, bias=False)
    bn = nn.BatchNorm2d(256)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(256, generator=g) + 0.5)
        bn.running_var.copy_(torch.rand(256, generator=g) + 0.5)
    import copy

    conv_d, bn_d = copy.deepcopy(conv).cuda(), copy.deepcopy(bn).cuda()
    bn.train(train)
    bn_d.train(train)
    xr = x.clone().requires_grad_(True)
    ref = bn(F.interpolate(F.relu(conv(F.adaptive_avg_pool2d(xr, 1))), size=(9, 9), mode="bilinear", align_corners=True))
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    xd = _cl(x).requires_grad_(True)
    out = ops.broadcast_bn(ops.conv_bn_act(ops.global_avgpool(xd), conv_d, None, ops.ACT_RELU), bn_d, 9, 9)
    _close(out, ref, 2e-4, "pool branch fwd")
    out.backward(_cl(go))
    _close(xd.grad, xr.grad, 1e-3, "pool branch dx")
    _close(conv_d.weight.grad, conv.weight.grad, 1e-3, "pool branch dw")
    _close(bn_d.weight.grad, bn.weight.grad, 1e-3, "pool branch dgamma")
    if train:
        _close(bn_d.running_var, bn.running_var, 1e-4, "pool branch running_var")


def test_dropout_channel_scale():
    ops = _ops()
    g = torch.Generator().manual_seed(8)
    x = torch.randn(2, 256, 7, 7, generator=g)
    mask = (torch.rand(2, 256, generator=g) > 0.5).float() * 2.0
    conv = nn.Conv2d(256, 256, 1, bias=False)
    bn = nn.BatchNorm2d(256)
    import copy

    conv_d, bn_d = copy.deepcopy(conv).cuda(), copy.deepcopy(bn).cuda()
    xr = x.clone().requires_grad_(True)
    ref = F.relu(bn(conv(xr))) * mask[:, :, None, None]
    go = torch.randn(ref.shape, generator=g)
    ref.backward(go)
    xd = _cl(x).requires_grad_(True)
    out = ops.conv_bn_act(xd, conv_d, bn_d, ops.ACT_RELU, nc_scale=mask.cuda())
    _close(out, ref, 2e-4, "dropout fwd")
    out.backward(_cl(go))
    _close(xd.grad, xr.grad, 5e-4, "dropout dx")
    _close(bn_d.weight.grad, bn.weight.grad, 5e-4, "dropout dgamma")


@pytest.mark.parametrize("weighted", [False, True])
def test_cross_entropy(weighted):
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    logit = torch.randn(2, 19, 33, 29, generator=g) * 3
    target = torch.randint(0, 19, (2, 33, 29), generator=g).float()
    target[:, :4] = 255
    wt = (torch.rand(19, generator=g) + 0.5) if weighted else None
    lr = logit.clone().requires_grad_(True)
    ref = F.cross_entropy(lr, target.long(), weight=wt, ignore_index=255) / 2
    ref.backward()
    ld = logit.cuda().requires_grad_(True)
    out = ops.cross_entropy(ld, target.cuda(), wt.cuda() if weighted else None, 255) / 2
    assert abs(out.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
    out.backward()
    _close(ld.grad, lr.grad, 1e-4, "ce grad")
    # int64 targets too
    out2 = ops.cross_entropy(logit.cuda(), target.long().cuda(), wt.cuda() if weighted else None, 255) / 2
    assert abs(out2.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))


def test_scoring_kernels():
    ops = _ops()
    g = torch.Generator().manual_seed(10)
    n, t, h, w, c = 2, 10, 37, 41, 19
    votes = torch.randint(0, c, (n, t, h, w), generator=g).to(torch.uint8)
    votes[:, :, :10] = votes[:, :1, :10]  # unanimous rows -> zero entropy
    label = torch.randint(0, c, (n, h, w), generator=g).float()
    label[:, :3] = 255
    label[:, 3:4] = -1
    ent = torch.zeros(n, h, w)
    vf = votes.float()
    for i in range(n):
        e = torch.zeros(h, w)
        for cc in range(c):
            p = torch.sum(vf[i] == cc, dim=0, dtype=torch.float32) / t
            e = e - p * torch.log2(p + 1e-12)
        e[(label[i] < 0) | (label[i] >= c)] = 0
        ent[i] = e
    emap, mean = ops.vote_entropy(votes.cuda(), label.cuda(), c)
    _close(emap, ent, 1e-5, "vote entropy map")
    _close(mean, ent.mean(dim=(1, 2)), 1e-5, "vote entropy mean")
    logits = torch.randn(n, c, h, w, generator=g) * 2
    sm = torch.softmax(logits, 1)
    mask = (label < 0) | (label >= c)
    conf = sm.max(1)[0].clone()
    conf[mask] = 1
    top2 = sm.topk(2, dim=1)[0]
    margin = (top2[:, 0] - top2[:, 1]).clone()
    margin[mask] = 1
    sent = -(sm * torch.log2(sm + 1e-12)).sum(1)
    sent[mask] = 0
    for mode, ref in ((0, conf), (1, margin), (2, sent)):
        smap, smean = ops.softmax_scores(logits.cuda(), label.cuda(), c, mode, want_map=True)
        _close(smap, ref, 1e-5, "softmax score map mode %d" % mode)
        _close(smean, ref.mean(dim=(1, 2)), 1e-5, "softmax score mean mode %d" % mode)
    wl = ops.weak_labels(logits.cuda(), label.cuda(), c).cpu()
    ref_wl = logits.argmax(1).to(torch.uint8)
    ref_wl[mask] = 255
    assert torch.equal(wl, ref_wl)
    # fused upsample + argmax vs interpolate + argmax (exact wherever the top-2 margin is not tiny)
    low = torch.randn(n, c, 10, 11, generator=g)
    up = F.interpolate(low, size=(h, w), mode="bilinear", align_corners=True)
    top = up.topk(2, dim=1)[0]
    safe = (top[:, 0] - top[:, 1]) > 1e-4
    lowd = torch.zeros(n, 10, 11, 20).cuda()
    lowd[..., :c] = low.permute(0, 2, 3, 1).cuda()
    lowd = lowd.permute(0, 3, 1, 2)[:, :c]
    vt = torch.zeros(n, 1, h, w, dtype=torch.uint8).cuda()
    ops.upsample_argmax(lowd, h, w, vt, 0)
    got = vt[:, 0].cpu().long()
    assert torch.equal(got[safe], up.argmax(1)[safe])
    assert safe.float().mean() > 0.99


def test_coreset_kernels():
    ops = _ops()
    g = torch.Generator().manual_seed(12)
    feat = torch.randn(2, 304, 129, 129, generator=g)
    ref = F.avg_pool2d(feat, (64, 64), 32).flatten(1)
    got = ops.avgpool_features(_cl(feat), 64, 32)
    _close(got, ref, 1e-5, "avgpool features")
    assert got.shape[1] == 2736
    import numpy as np

    pts = np.array([[0, 0], [0, 1], [1, 1], [1, 0], [10, 10], [11, 10], [10, 11], [20, 20], [21, 20]], dtype=np.float32)
    # brute-force greedy reference (core_set.py:17-38 semantics)
    def greedy(f, sel, k):
        d = np.min(np.linalg.norm(f[:, None, :].astype(np.float64) - f[sel][None].astype(np.float64), axis=2), axis=1)
        out = []
        for _ in range(k):
            i = int(np.argmax(d))
            out.append(i)
            d = np.minimum(d, np.linalg.norm(f.astype(np.float64) - f[i].astype(np.float64), axis=1))
        return out

    picks, _ = ops.kcenter_greedy(torch.from_numpy(pts).cuda(), [6], 5)
    assert picks.cpu().tolist() == greedy(pts, [6], 5)
    f = torch.randn(300, 2736, generator=g).numpy()
    picks, _ = ops.kcenter_greedy(torch.from_numpy(f).cuda(), list(range(10)), 20)
    assert picks.cpu().tolist() == greedy(f, list(range(10)), 20)


def test_region_kernels():
    ops = _ops()
    g = torch.Generator().manual_seed(13)
    maps = torch.rand(3, 65, 65, generator=g)
    r = 17
    ref = F.conv2d(maps[:, None], torch.ones(1, 1, r, r))[:, 0]
    got = ops.box_sum(maps.cuda(), r)
    _close(got, ref, 1e-5, "box sum")
    lo, hi = ref.min(), ref.max()
    refn = ref.clone().add_(-lo).mul_(1.0 / (hi - lo))
    ops.minmax_normalize_(got)
    _close(got, refn, 1e-5, "minmax")
    # NMS vs a line-by-line CPU loop of mc_dropout.py:82-108
    sm = refn.clone()
    regions = [[] for _ in range(sm.shape[0])]
    cnt = 0
    for _ in range(12):
        am = sm.view(-1).argmax()
        i, rr, cc = am // (sm.shape[1] * sm.shape[2]), (am // sm.shape[2]) % sm.shape[1], am % sm.shape[2]
        regions[i.item()].append((rr.item(), cc.item(), r, r))
        cnt += 1
        r0, c0 = max(0, rr - r), max(0, cc - r)
        r1, c1 = min(sm.shape[1], rr + r), min(sm.shape[2], cc + r)
        sm[i, r0:r1, c0:c1] = 0
        if sm.max() < 0.01:
            break
    got_regions, got_cnt = ops.square_nms(refn.clone().cuda(), r, 12)
    assert got_cnt == cnt and got_regions == regions


def test_weight_split_operand_is_exact():
    """DASS_F32X6 operand (dass_weight_transform / dass_weight_split_batch): every weight equals the sum of its three
    bf16 parts to <= 2^-24 |w| (measured: exact for almost all values), for forward and dgrad layouts, including zero
    padding of the 32-wide slabs, tiny and huge magnitudes."""
    ops = _ops()
    ops.set_f32_mma("bf16x6")
    k, r, s, c = 40, 3, 3, 36  # neither multiple of 32: slabs are zero padded
    g = torch.Generator().manual_seed(5)
    w = torch.randn(k, r, s, c, generator=g)
    w.view(-1)[:64] *= 1e-30
    w.view(-1)[64:128] *= 1e30
    w.view(-1)[128:136] = 0.0
    wd = w.cuda()
    for mode in (0, 1):
        rows, red = (k, c) if mode == 0 else (c, k)
        cch = (red + 31) // 32
        op = ops.prepare_conv_weight(wd, mode)
        assert op.numel() == rows * r * s * cch * 192 + 16   # (+ the trailer every pre-split operand ends in)
        parts = op[:-16].view(torch.bfloat16).view(rows, r * s, cch, 3, 32).double().cpu()
        recon = parts.sum(3).reshape(rows, r, s, cch * 32)
        if mode == 0:
            ref = torch.zeros(rows, r, s, cch * 32, dtype=torch.float64)
            ref[..., :c] = w.double()
        else:  # [c][r][s][k] with taps flipped
            ref = torch.zeros(rows, r, s, cch * 32, dtype=torch.float64)
            ref[..., :k] = w.double().permute(3, 1, 2, 0).flip(1, 2)
        err = (recon - ref).abs()
        assert (err <= ref.abs() * 2.0 ** -24).all(), err.max()
        assert (recon[..., red:] == 0).all()


def test_sgd_multi_matches_torch_sgd():
    """dass_hip.optim.SGD (one multi-tensor kernel) against torch.optim.SGD over three steps: two lr groups, momentum,
    weight decay, channels_last conv weights, tensors that are not a multiple of the 2048-element blocks, a parameter
    without gradient, and state_dict interchange with the stock optimizer."""
    _ops()
    from dass_hip.optim import SGD

    def make():
        torch.manual_seed(4)
        ws = [torch.randn(40, 36, 3, 3).cuda().contiguous(memory_format=torch.channels_last), torch.randn(5000).cuda(), torch.randn(7).cuda(),
              torch.randn(64, 64, 1, 1).cuda().contiguous(memory_format=torch.channels_last), torch.randn(3).cuda()]
        return [torch.nn.Parameter(w) for w in ws]

    pa, pb = make(), make()
    oa = torch.optim.SGD([{"params": pa[:2], "lr": 0.01}, {"params": pa[2:], "lr": 0.1}], momentum=0.9, weight_decay=5e-4)
    ob = SGD([{"params": pb[:2], "lr": 0.01}, {"params": pb[2:], "lr": 0.1}], momentum=0.9, weight_decay=5e-4)
    g = torch.Generator(device="cuda").manual_seed(9)
    for step in range(3):
        for a, b in zip(pa[:4], pb[:4]):  # the last parameter never gets a gradient
            gr = torch.randn(a.shape, device="cuda", generator=g).contiguous(memory_format=torch.channels_last if a.dim() == 4 else torch.contiguous_format)
            a.grad, b.grad = gr.clone(memory_format=torch.preserve_format), gr.clone(memory_format=torch.preserve_format)
        oa.step()
        ob.step()
        for i, (a, b) in enumerate(zip(pa, pb)):
            assert (a - b).abs().max().item() <= 2e-6 * max(1.0, a.abs().max().item()), (step, i)
    assert all("momentum_buffer" in ob.state[q] for q in pb[:4])
    sd = ob.state_dict()
    oc = torch.optim.SGD([{"params": pa[:2], "lr": 0.01}, {"params": pa[2:], "lr": 0.1}], momentum=0.9, weight_decay=5e-4)
    oc.load_state_dict(sd)  # same format as the stock optimizer's
    assert torch.allclose(oc.state[pa[0]]["momentum_buffer"], ob.state[pb[0]]["momentum_buffer"])


def test_conv_fuzz_random_shapes():
    """tools/conv_fuzz.py: 60 random (channels, size, kernel, stride, padding, dilation) x (BN train/eval, bias, residual,
    activation) cases in both parity engines against an f64 torch reference -- forward within 2e-5 rel-L2, every
    gradient within the ReLU-kink scale"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "conv_fuzz.py"), "60", "3"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "CASE" not in out.stdout, out.stdout[-3000:]
    assert "worst rel-L2" in out.stdout


def test_scoring_kernels_random_sizes():
    """vote entropy and the three CEAL softmax scores on 12 random (N, T, C, H, W) against the oracle's reductions: ragged
    sizes (non-multiples of every block dimension), T up to 40, C up to 60, ignore labels, single-pixel rows"""
    import random

    from oracle import selection_cpu as S

    ops = _ops()
    rng = random.Random(5)
    for case in range(12):
        n, t, c = rng.randint(1, 3), rng.choice([1, 2, 7, 10, 20, 40]), rng.choice([2, 4, 19, 21, 60])
        h, w = rng.choice([1, 3, 17, 64, 129]), rng.choice([1, 5, 33, 100, 257])
        g = torch.Generator().manual_seed(100 + case)
        votes = torch.randint(0, c, (n, t, h, w), generator=g).to(torch.uint8)
        label = torch.randint(0, c, (n, h, w), generator=g).float()
        label[torch.rand(n, h, w, generator=g) < 0.1] = 255
        emap, mean = ops.vote_entropy(votes.cuda(), label.cuda(), c)
        ref = S.vote_entropy_maps(votes.long(), label, c)
        ref = torch.stack([torch.as_tensor(r) for r in ref]).float() if not torch.is_tensor(ref) else ref.float()
        assert (emap.cpu() - ref).abs().max().item() <= 2e-5, (case, n, t, c, h, w)
        assert (mean.cpu() - ref.mean(dim=(1, 2))).abs().max().item() <= 2e-5
        logits = torch.randn(n, c, h, w, generator=g) * 3
        maps = S.softmax_score_maps(logits, label, c)
        for mode in range(3):
            smap, smean = ops.softmax_scores(logits.cuda(), label.cuda(), c, mode, want_map=True)
            r = maps[mode] if isinstance(maps, (list, tuple)) else maps[:, mode]
            r = torch.as_tensor(r).float()
            assert (smap.cpu() - r).abs().max().item() <= 3e-5, (case, mode, n, t, c, h, w)
            assert (smean.cpu() - r.mean(dim=(1, 2))).abs().max().item() <= 3e-5


@pytest.mark.parametrize("shape", [(2, 19, 33, 33, 129, 129), (1, 21, 25, 25, 97, 97), (3, 19, 10, 11, 37, 41), (1, 19, 129, 129, 513, 513),
                                   (2, 5, 7, 9, 25, 30)])
def test_upsample_argmax_vector_kernel_equals_scalar_kernel(shape):
    """dass_upsample_argmax takes a four-pixels-per-thread path (16-byte class-vector loads) when the rows are 16-byte
    aligned and the upsampling factor is >= 3; a 21-float row pitch forces the one-pixel-per-thread kernel.  Same
    arithmetic per pixel, same first-maximum rule: the votes must be identical, ragged right edge included."""
    ops = _ops()
    n, c, ih, iw, oh, ow = shape
    g = torch.Generator().manual_seed(c * ih + ow)
    low = torch.randn(n, ih, iw, c, generator=g)
    low[:, ::3, ::2, 1] = low[:, ::3, ::2, 0]  # exact ties between classes 0 and 1 on a lattice of corners
    votes = []
    for pitch in (24, 21):
        buf = torch.zeros(n, ih, iw, pitch)
        buf[..., :c] = low
        x = buf.cuda().permute(0, 3, 1, 2)[:, :c]
        vt = torch.full((n, 1, oh, ow), 255, dtype=torch.uint8).cuda()
        ops.upsample_argmax(x, oh, ow, vt, 0)
        votes.append(vt.cpu())
    assert torch.equal(votes[0], votes[1])
    assert int(votes[0].max()) < c
