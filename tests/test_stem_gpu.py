"""The stems' f32-MFMA kernels (csrc/stem_rowtap.hip) behind dass_conv2d_rowtap / dass_conv2d_rowtap_wgrad, through the C-ABI,
against f64 convolutions of the same inputs (reference sites: models/backbone/resnet.py:65, mobilenet.py:14).  Tolerance: f32
products and f32 accumulation of <= 147 terms (forward) / <= 2.1 M terms (weight gradient, summed in blocks) -- 2e-5 relative to the
largest output, stated below; the generic implicit-GEMM kernels (DASS_ROWTAP_FAST=0) are held to the same bound."""
import ctypes
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # n, h, w, cin, k, r, stride, pad
    (2, 65, 65, 3, 64, 7, 2, 3),     # R101 / R50 stem, odd image
    (1, 33, 47, 3, 64, 7, 2, 3),     # ragged: 17 x 24 = 408 output pixels (not a multiple of 32), every border case
    (3, 64, 50, 3, 32, 3, 2, 1),     # MobileNetV2 / Xception stem
    (8, 129, 129, 3, 64, 7, 2, 3),
]


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _run(fast, n, h, w, cin, k, r, stride, pad, seed=0):
    from dass_hip._lib import lib

    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, h, w, cin, generator=g)
    wt = torch.randn(k, r, r, cin, generator=g) * 0.1
    oh, ow = (h + 2 * pad - r) // stride + 1, (w + 2 * pad - r) // stride + 1
    dy = torch.randn(n, oh, ow, k, generator=g)
    xd, wd, dyd = x.cuda(), wt.cuda(), dy.cuda()
    y = torch.full((n, oh, ow, k), float("nan"), device="cuda")
    dw = torch.full((k, r, r, cin), float("nan"), device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    keep = os.environ.get("DASS_ROWTAP_FAST")
    os.environ["DASS_ROWTAP_FAST"] = "1" if fast else "0"
    try:
        assert lib.dass_conv2d_rowtap(_p(xd), _p(wd), _p(y), k, n, h, w, cin, oh, ow, k, r, r, stride, pad, 0, st) == 0
        assert lib.dass_conv2d_rowtap_wgrad(_p(xd), _p(dyd), k, _p(dw), n, h, w, cin, oh, ow, k, r, r, stride, pad, 0, st) == 0
        torch.cuda.synchronize()
    finally:
        if keep is None:
            os.environ.pop("DASS_ROWTAP_FAST", None)
        else:
            os.environ["DASS_ROWTAP_FAST"] = keep
    x64 = x.double().permute(0, 3, 1, 2).requires_grad_(False)
    w64 = wt.double().permute(0, 3, 1, 2).clone().requires_grad_(True)
    y64 = torch.nn.functional.conv2d(x64, w64, stride=stride, padding=pad)
    (y64 * dy.double().permute(0, 3, 1, 2)).sum().backward()
    return y.cpu().double().permute(0, 3, 1, 2), y64.detach(), dw.cpu().double().permute(0, 3, 1, 2), w64.grad


@pytest.mark.parametrize("fast", [True, False])
@pytest.mark.parametrize("shape", SHAPES)
def test_stem_forward_and_weight_gradient_vs_f64(shape, fast):
    y, y64, dw, dw64 = _run(fast, *shape)
    assert torch.isfinite(y).all() and torch.isfinite(dw).all()
    ey = (y - y64).abs().max().item() / y64.abs().max().item()
    ew = (dw - dw64).abs().max().item() / dw64.abs().max().item()
    print(shape, "fast" if fast else "generic", "forward %.2e  weight gradient %.2e" % (ey, ew))
    tol = 2e-5 if fast else 2e-4   # (the generic forward multiplies bf16 splits of the f32 values)
    assert ey <= tol and ew <= 2e-5


def test_stem_full_size_timing():
    """R101 stem at the headline shape (8 x 513^2): the specialised kernels against the generic ones, same inputs, and their times"""
    from dass_hip._lib import lib

    n, h, w, cin, k, r, stride, pad = 8, 513, 513, 3, 64, 7, 2, 3
    oh = ow = 257
    g = torch.Generator().manual_seed(5)
    xd = torch.randn(n, h, w, cin, generator=g).cuda()
    wd = (torch.randn(k, r, r, cin, generator=g) * 0.1).cuda()
    dyd = torch.randn(n, oh, ow, k, generator=g).cuda()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}
    keep = os.environ.get("DASS_ROWTAP_FAST")
    try:
        for fast in ("1", "0"):
            os.environ["DASS_ROWTAP_FAST"] = fast
            y = torch.empty((n, oh, ow, k), device="cuda")
            dw = torch.empty((k, r, r, cin), device="cuda")
            times = []
            for which in ("fwd", "wgrad"):
                for it in range(6):
                    if it == 1:
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                    if which == "fwd":
                        assert lib.dass_conv2d_rowtap(_p(xd), _p(wd), _p(y), k, n, h, w, cin, oh, ow, k, r, r, stride, pad, 0, st) == 0
                    else:
                        assert lib.dass_conv2d_rowtap_wgrad(_p(xd), _p(dyd), k, _p(dw), n, h, w, cin, oh, ow, k, r, r, stride, pad, 0, st) == 0
                torch.cuda.synchronize()
                times.append((time.perf_counter() - t0) / 5 * 1e6)
            out[fast] = (y.clone(), dw.clone(), times)
    finally:
        if keep is None:
            os.environ.pop("DASS_ROWTAP_FAST", None)
        else:
            os.environ["DASS_ROWTAP_FAST"] = keep
    print("stem 8x513^2: specialised fwd %.0f us, wgrad %.0f us; generic fwd %.0f us, wgrad %.0f us" % (*out["1"][2], *out["0"][2]))
    ey = (out["1"][0] - out["0"][0]).abs().max().item() / out["0"][0].abs().max().item()
    ew = (out["1"][1] - out["0"][1]).abs().max().item() / out["0"][1].abs().max().item()
    assert ey <= 2e-4 and ew <= 2e-5, (ey, ew)


DW_SHAPES = [  # n, h, w, c, stride, dil
    (2, 33, 33, 96, 1, 1),
    (2, 65, 47, 144, 2, 1),     # ragged channels (144 = 2.25 x 64), stride 2, output rows of 24 pixels (1.5 strips)
    (3, 33, 33, 576, 1, 2),     # the dilated blocks of MobileNetV2 at os16
    (1, 17, 19, 32, 2, 2),
    (16, 129, 129, 144, 1, 1),  # config C size
]


@pytest.mark.parametrize("strip", ["1", "0"])
@pytest.mark.parametrize("shape", DW_SHAPES)
def test_depthwise_weight_gradient_strip_kernel_vs_f64(shape, strip):
    """dass_dwconv3x3_bwd_weight: the strip kernel (window in registers, csrc/dwconv_region.hip) and the pixel-cursor kernel it replaces
    (DASS_DW_WGRAD_STRIP=0) against an f64 convolution's weight gradient; f32 sums of up to 266 k terms: 2e-5 of the largest value"""
    from dass_hip._lib import lib

    n, h, w, c, stride, dil = shape
    pad = dil
    oh, ow = (h + 2 * pad - 2 * dil - 1) // stride + 1, (w + 2 * pad - 2 * dil - 1) // stride + 1
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, h, w, c, generator=g)
    dy = torch.randn(n, oh, ow, c, generator=g)
    xd, dyd = x.cuda(), dy.cuda()
    dw = torch.full((c, 9), float("nan"), device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    keep = os.environ.get("DASS_DW_WGRAD_STRIP")
    os.environ["DASS_DW_WGRAD_STRIP"] = strip
    try:
        t = []
        for it in range(4):
            if it == 1:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            assert lib.dass_dwconv3x3_bwd_weight(_p(xd), c, _p(dyd), c, _p(dw), n, h, w, c, oh, ow, stride, pad, dil, 0, st) == 0
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 3 * 1e6
    finally:
        if keep is None:
            os.environ.pop("DASS_DW_WGRAD_STRIP", None)
        else:
            os.environ["DASS_DW_WGRAD_STRIP"] = keep
    w64 = torch.zeros(c, 1, 3, 3, dtype=torch.float64, requires_grad=True)
    y64 = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2), w64, stride=stride, padding=pad, dilation=dil, groups=c)
    (y64 * dy.double().permute(0, 3, 1, 2)).sum().backward()
    ref = w64.grad.reshape(c, 9)
    err = (dw.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    gb = (x.numel() + dy.numel()) * 4 / 1e9
    print(shape, "strip" if strip == "1" else "cursor", "err %.2e  %.0f us  %.2f TB/s" % (err, us, gb / us * 1e6 / 1e3))
    assert err <= 2e-5


@pytest.mark.parametrize("shape", DW_SHAPES + [(2, 33, 33, 64, 1, 1), (1, 9, 40, 1024, 1, 2)])
def test_depthwise_forward_and_input_gradient_strip_kernels(shape):
    """dass_dwconv3x3_fwd / dass_dwconv3x3_bwd_data: the strip kernels against an f64 convolution (1e-5 of the largest value) and
    against the per-pixel kernels they replace (DASS_DW_STRIP=0): the products are added in the same order, so the bits are equal"""
    from dass_hip._lib import lib

    n, h, w, c, stride, dil = shape
    pad = dil
    oh, ow = (h + 2 * pad - 2 * dil - 1) // stride + 1, (w + 2 * pad - 2 * dil - 1) // stride + 1
    g = torch.Generator().manual_seed(4)
    x = torch.randn(n, h, w, c, generator=g)
    wt = torch.randn(c, 9, generator=g)
    dy = torch.randn(n, oh, ow, c, generator=g)
    xd, wd, dyd = x.cuda(), wt.cuda(), dy.cuda()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    res = {}
    keep = os.environ.get("DASS_DW_STRIP")
    try:
        for strip in ("1", "0"):
            os.environ["DASS_DW_STRIP"] = strip
            y = torch.full((n, oh, ow, c), float("nan"), device="cuda")
            dx = torch.full((n, h, w, c), float("nan"), device="cuda")
            us = []
            for fn in ("fwd", "bwd"):
                for it in range(4):
                    if it == 1:
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                    if fn == "fwd":
                        assert lib.dass_dwconv3x3_fwd(_p(xd), c, _p(wd), _p(y), c, n, h, w, c, oh, ow, stride, pad, dil, 0, st) == 0
                    else:
                        assert lib.dass_dwconv3x3_bwd_data(_p(dyd), c, _p(wd), _p(dx), c, n, h, w, c, oh, ow, stride, pad, dil, 0, st) == 0
                torch.cuda.synchronize()
                us.append((time.perf_counter() - t0) / 3 * 1e6)
            res[strip] = (y.cpu(), dx.cpu(), us)
    finally:
        if keep is None:
            os.environ.pop("DASS_DW_STRIP", None)
        else:
            os.environ["DASS_DW_STRIP"] = keep
    x64 = x.double().permute(0, 3, 1, 2).clone().requires_grad_(True)
    y64 = torch.nn.functional.conv2d(x64, wt.double().reshape(c, 1, 3, 3), stride=stride, padding=pad, dilation=dil, groups=c)
    (y64 * dy.double().permute(0, 3, 1, 2)).sum().backward()
    ey = (res["1"][0].double().permute(0, 3, 1, 2) - y64.detach()).abs().max().item() / y64.abs().max().item()
    ex = (res["1"][1].double().permute(0, 3, 1, 2) - x64.grad).abs().max().item() / x64.grad.abs().max().item()
    gb = (x.numel() + dy.numel()) * 4 / 1e3
    print(shape, "fwd %.1e (%.0f us, %.2f TB/s; per-pixel kernel %.0f us)  input gradient %.1e (%.0f us, %.2f TB/s; per-pixel %.0f us)" % (
        ey, res["1"][2][0], gb / res["1"][2][0] / 1e3, res["0"][2][0], ex, res["1"][2][1], gb / res["1"][2][1] / 1e3, res["0"][2][1]))
    assert ey <= 1e-5 and ex <= 1e-5
    assert torch.equal(res["1"][0], res["0"][0])
    assert torch.equal(res["1"][1], res["0"][1])


def test_depthwise_input_gradient_carries_the_expand_layers_bn_sums():
    """dass_dwconv3x3_bwd_data_bnstats: one train step of DeepLab-MobileNetV2 with the BN-backward sums of every expand 1x1 riding in the
    depthwise conv's input-gradient launch, against the same step with the separate reduce pass (ops.set_dw_bn_link(False)): same loss,
    every gradient within 2e-5 of its largest value (the sums are added in a different order), and the launches did carry them"""
    from dass_hip import ops
    from models.deeplab import DeepLab
    from oracle import deeplab_cpu as O
    from utils.loss import SegmentationLosses

    keep = ops.f32_mma()
    try:
        ops.set_compute_dtype(torch.float32)
        ops.set_f32_mma("f16x3")
        om = O.ODeepLab("mobilenet", 16, 21)
        O.fill_state_dict(om, seed=5, randomize_bn_stats=False)
        x, lab = O.synthetic_batch(4, 97, 97, 21, first_index=900)
        m1, m2 = O.dropout_masks(4, 1, seed=6)
        res = {}
        for on in (True, False, "fwd-off"):
            # (link on / off with the forward statistics fused in both; "fwd-off": the link on, dass_dwconv3x3_fwd_sums off)
            ops.set_dw_bn_link(on is not False, fwd_sums=on != "fwd-off")
            pm = DeepLab(backbone="mobilenet", output_stride=16, num_classes=21, sync_bn=False, pretrained=False)
            pm.load_state_dict(om.state_dict())
            pm = pm.cuda().train()
            before = dict(ops.bn_link_counts)
            loss = SegmentationLosses(cuda=True).build_loss("ce")(pm(x.cuda(), dropout_masks=(m1[0].cuda(), m2[0].cuda())), lab.cuda())
            loss.backward()
            torch.cuda.synchronize()
            res[on] = (loss.item(), {k: p.grad.double().cpu() for k, p in pm.named_parameters()},
                       {k: ops.bn_link_counts[k] - before[k] for k in before})
        print("link counts with the depthwise link:", res[True][2], "without:", res[False][2])
        assert res[True][0] == res[False][0]
        # forward statistics in the conv launch vs dass_channel_sums: another summation order of the same values; the loss moves in its last digits
        # (and the batch-4 BN of the ASPP image-pool branch turns that into percents of ITS gradients: not compared -- kernel-level test above)
        assert abs(res[True][0] - res["fwd-off"][0]) <= 2e-6 * abs(res[True][0])
        assert res[True][2]["used"] >= res[False][2]["used"] + 10   # MobileNetV2 at os16: 14 stride-1 depthwise convs behind an expand layer
        worst = max(((res[True][1][k] - res[False][1][k]).abs().max().item() / max(res[False][1][k].abs().max().item(), 1e-12), k) for k in res[True][1])
        print("largest gradient difference %.2e (%s)" % worst)
        assert worst[0] <= 2e-5, worst
    finally:
        ops.set_dw_bn_link(True, fwd_sums=True)
        ops.set_f32_mma(keep)


@pytest.mark.parametrize("shape", [(2, 33, 33, 96, 1, 1), (3, 33, 33, 576, 1, 2), (4, 49, 49, 32, 1, 1), (2, 17, 40, 144, 1, 1)])
@pytest.mark.parametrize("act", [0, 2])
def test_depthwise_bnstats_kernel_vs_f64(shape, act):
    """dass_dwconv3x3_bwd_data_bnstats at kernel level: dx equals dass_dwconv3x3_bwd_data bit for bit, and the three per-channel sums
    (sum dz, sum dz xhat, max |dz|) equal an f64 evaluation of the same dx to 1e-5 of their largest value (act 2 = ReLU6 gate from
    fma(y, scale, shift), act 0 = no activation)"""
    from dass_hip._lib import lib

    n, h, w, c, stride, dil = shape
    pad = dil
    g = torch.Generator().manual_seed(11)
    dy = torch.randn(n, h, w, c, generator=g)
    wt = torch.randn(c, 9, generator=g)
    y = torch.randn(n, h, w, c, generator=g) * 2 + 1
    mean, invstd = torch.randn(c, generator=g) * 0.1 + 1, torch.rand(c, generator=g) + 0.5
    sc, sh = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    d = lambda t: t.cuda()
    dyd, wd, yd, md, isd, scd, shd = d(dy), d(wt), d(y), d(mean), d(invstd), d(sc), d(sh)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    dx0 = torch.empty((n, h, w, c), device="cuda")
    dx1 = torch.empty((n, h, w, c), device="cuda")
    sums = torch.zeros((2 * c + (c + 1) // 2,), dtype=torch.float64, device="cuda")
    assert lib.dass_dwconv3x3_bwd_data(_p(dyd), c, _p(wd), _p(dx0), c, n, h, w, c, h, w, stride, pad, dil, 0, st) == 0
    assert lib.dass_dwconv3x3_bwd_data_bnstats(_p(dyd), c, _p(wd), _p(dx1), c, n, h, w, c, h, w, stride, pad, dil, _p(yd), _p(md), _p(isd),
                                               _p(scd), _p(shd), act, _p(sums), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(dx0, dx1)
    dxr = dx0.cpu().double().reshape(-1, c)
    yr = y.double().reshape(-1, c)
    o = torch.from_numpy(np.float32(yr.numpy())).float() * sc + sh   # (the gate is taken from the f32 fma, as the forward computed it)
    o = torch.addcmul(sh, y.reshape(-1, c), sc)
    gate = ((o > 0) & (o < 6)).double() if act == 2 else torch.ones_like(yr)
    dz = dxr * gate
    xhat = (yr - mean.double()) * invstd.double()
    ref = torch.stack([dz.sum(0), (dz * xhat).sum(0)])
    got = sums[:2 * c].reshape(2, c).cpu()
    mx = sums[2 * c:].view(torch.float32)[:c].cpu()
    for i, name in enumerate(("sum dz", "sum dz xhat")):
        err = (got[i] - ref[i]).abs().max().item() / ref[i].abs().max().item()
        print(shape, act, name, "%.2e" % err)
        assert err <= 1e-5, (name, err)
    assert (mx.double() - dz.abs().max(0).values).abs().max().item() <= 1e-6 * dz.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 33, 33, 96, 1, 1), (2, 65, 47, 144, 2, 1), (3, 33, 33, 576, 1, 2), (4, 49, 49, 32, 1, 1)])
def test_depthwise_forward_sums_kernel_vs_f64(shape):
    """dass_dwconv3x3_fwd_sums: y equals dass_dwconv3x3_fwd bit for bit; the per-channel sum and sum of squares equal an f64 evaluation
    of that y to 1e-5 of their largest value"""
    from dass_hip._lib import lib

    n, h, w, c, stride, dil = shape
    pad = dil
    oh, ow = (h + 2 * pad - 2 * dil - 1) // stride + 1, (w + 2 * pad - 2 * dil - 1) // stride + 1
    g = torch.Generator().manual_seed(12)
    xd = (torch.randn(n, h, w, c, generator=g) + 0.3).cuda()
    wd = torch.randn(c, 9, generator=g).cuda()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    y0 = torch.empty((n, oh, ow, c), device="cuda")
    y1 = torch.empty((n, oh, ow, c), device="cuda")
    sums = torch.zeros((2 * c,), dtype=torch.float64, device="cuda")
    assert lib.dass_dwconv3x3_fwd(_p(xd), c, _p(wd), _p(y0), c, n, h, w, c, oh, ow, stride, pad, dil, 0, st) == 0
    assert lib.dass_dwconv3x3_fwd_sums(_p(xd), c, _p(wd), _p(y1), c, n, h, w, c, oh, ow, stride, pad, dil, _p(sums), st) == 0
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    yr = y0.cpu().double().reshape(-1, c)
    ref = torch.stack([yr.sum(0), (yr * yr).sum(0)])
    got = sums.reshape(2, c).cpu()
    for i in range(2):
        assert (got[i] - ref[i]).abs().max().item() <= 1e-5 * ref[i].abs().max().item(), i


@pytest.mark.parametrize("shape", [(2, 19, 33, 33), (1, 21, 129, 129), (3, 4, 5, 9), (2, 19, 2, 2)])
def test_bilinear_backward_ratio4_kernel_equals_general_gather(shape):
    """dass_bilinear_bwd, NCHW gradient of the x4 align_corners upsampling of the logits (deeplab.py:45: 129 -> 513): the constant-weight
    kernel against the general gather (DASS_BILINEAR_R4=0) to the last bits (1e-6), and against autograd of F.interpolate in f64 (1e-5 of the largest value)"""
    from dass_hip._lib import lib

    n, c, ih, iw = shape
    oh, ow = 4 * (ih - 1) + 1, 4 * (iw - 1) + 1
    g = torch.Generator().manual_seed(21)
    dy = torch.randn(n, c, oh, ow, generator=g)
    dyd = dy.cuda()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    cp = (c + 3) // 4 * 4
    out = {}
    keep = os.environ.get("DASS_BILINEAR_R4")
    try:
        for mode in ("1", "0"):
            os.environ["DASS_BILINEAR_R4"] = mode
            dx = torch.zeros((n, ih, iw, cp), device="cuda")
            assert lib.dass_bilinear_bwd(_p(dyd), 0, _p(dx), cp, n, ih, iw, c, oh, ow, 1, 0, st) == 0
            torch.cuda.synchronize()
            out[mode] = dx.cpu()
    finally:
        if keep is None:
            os.environ.pop("DASS_BILINEAR_R4", None)
        else:
            os.environ["DASS_BILINEAR_R4"] = keep
    # (same products in the same order; where the compiler contracts a multiply-add into an fma differs between the two kernels: last-bit differences)
    assert (out["1"] - out["0"]).abs().max().item() <= 1e-6 * out["0"].abs().max().item()
    x = torch.zeros(n, c, ih, iw, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.interpolate(x, size=(oh, ow), mode="bilinear", align_corners=True).backward(dy.double())
    got = out["1"][..., :c].permute(0, 3, 1, 2).double()
    assert (got - x.grad).abs().max().item() <= 1e-5 * x.grad.abs().max().item()
