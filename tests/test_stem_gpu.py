"""The stems' f32-MFMA kernels (csrc/stem_rowtap.hip) behind dass_conv2d_rowtap / dass_conv2d_rowtap_wgrad, through the C-ABI,
against f64 convolutions of the same inputs (reference sites: models/backbone/resnet.py:65, mobilenet.py:14).  Tolerance: f32
products and f32 accumulation of <= 147 terms (forward) / <= 2.1 M terms (weight gradient, summed in blocks) -- 2e-5 relative to the
largest output, stated below; the generic implicit-GEMM kernels (DASS_ROWTAP_FAST=0) are held to the same bound."""
import ctypes
import os
import time

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
