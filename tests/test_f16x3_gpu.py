"""The "f16x3" engine (pre-split kernels in their two-part mode, include/dass_hip.h "dass_set_x3_parts"): every f32 operand tensor
is scaled by a per-tensor power of two and split into two f16 parts (23 significant bits), three products per pair on the f16
MFMA pipe.  Checked against f64 convolutions computed outside the kernels, side by side with the plain f32-MFMA engine and the
six-product bf16 engine on the SAME inputs -- including the input classes that could break a scaled f16 format: tiny gradients
(1e-7), large activations (1e4), heavy tails (amax / rms ~ 1e3), all-zero tensors, a single huge outlier -- and end to end:
one train step of every backbone against the bf16x6 engine and against the CPU oracle, MC-dropout votes against the oracle.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (N, C, H, W, K, ksize, stride, pad, dil)
CASES = [(8, 256, 33, 33, 256, 3, 1, 1, 1), (2, 304, 33, 33, 256, 3, 1, 1, 1), (2, 256, 33, 33, 256, 3, 1, 6, 6),
         (1, 512, 33, 33, 256, 3, 1, 18, 18), (3, 1024, 17, 17, 256, 1, 1, 0, 1), (2, 64, 31, 29, 72, 1, 1, 0, 1),
         (2, 128, 35, 35, 128, 3, 2, 1, 1), (2, 48, 19, 23, 40, 3, 1, 1, 1), (1, 96, 9, 9, 320, 1, 1, 2, 1)]
TILES = [0, 11, 12, 13, 14, 15, 16, 17, 18, 21, 22, 24, 28]


@pytest.fixture(autouse=True)
def _engine():
    from dass_hip import ops
    from dass_hip._lib import lib

    mode, dt = ops.f32_mma(), ops.compute_dtype()
    ops.set_compute_dtype(torch.float32)
    ops.set_f32_mma("f16x3")
    yield
    lib.dass_x3_force_tile(0)
    ops.set_f32_mma(mode)
    ops.set_compute_dtype(dt)


def _inputs(case, kind="gauss"):
    n, c, h, wd, k, ks, stride, pad, dil = case
    g = torch.Generator().manual_seed(c * 7 + k)
    x = torch.randn(n, c, h, wd, generator=g)
    w = torch.randn(k, c, ks, ks, generator=g) * (2.0 / (c * ks * ks)) ** 0.5
    if kind == "tiny":          # gradient-like magnitudes
        x = x * 1e-7
    elif kind == "large":
        x = x * 1e4
    elif kind == "heavy":       # heavy tails: lognormal multipliers, amax / rms in the hundreds
        x = x * torch.exp(2.0 * torch.randn(n, c, h, wd, generator=g))
    elif kind == "outlier":     # one element 1e6 times the rest: everything else sits 20 binades below the bound
        x[0, 0, 0, 0] = 1e6
    elif kind == "relu":
        x = torch.relu(x)
    return x, w


def _rows(t_nchw):
    return t_nchw.permute(0, 2, 3, 1).contiguous().cuda()


def _rel(a, ref):
    return (a.double().cpu() - ref).norm().item() / max(ref.norm().item(), 1e-300)


def _fwd(ops, engine, case, x, w):
    n, c, h, wd, k, ks, stride, pad, dil = case
    ops.set_f32_mma(engine)
    oh, ow = ops.conv_out_size(h, ks, stride, pad, dil), ops.conv_out_size(wd, ks, stride, pad, dil)
    dims = (n, h, wd, c, oh, ow, k, ks, ks, stride, pad, dil)
    y = torch.full((n, oh, ow, k), float("nan"), device="cuda")
    xr = _rows(x)
    wk = w.permute(0, 2, 3, 1).contiguous().cuda()
    if engine == "f32":
        ops.conv_launch(xr, c, ops.prepare_conv_weight(wk), y, k, dims)
    else:
        ops.conv_x3_launch(ops.split3_rows(xr, c, n * h * wd, c), ops.prepare_conv_weight(wk, x3=True), y, k, dims)
    return y


def test_two_part_rows_carry_23_bits():
    """decode the operand a conv would read: (h0 + h1) * inv_scale equals x to 2^-22 |x| (elements within 2^10 of the bound) and
    to 2^-25 / scale absolutely (the rest); zero row and ragged channel tail are zero; the trailer holds 1 / scale and max |x|"""
    from dass_hip import ops

    g = torch.Generator().manual_seed(3)
    x = torch.randn(37, 72, generator=g) * torch.logspace(-3, 2, 72)[None]
    buf = ops.split3_rows(x.cuda(), 72, 37, 72)
    cc = 3
    assert buf.numel() == 38 * cc * 128 + 16
    tr = buf[-16:].view(torch.float32).cpu()
    inv, bound = float(tr[0]), float(tr[1])
    assert bound == float(x.abs().max()) and 2 ** 14 <= bound / inv < 2 ** 15
    v = buf[:-16].view(torch.float16).view(38, cc, 2, 32).cpu().double()
    back = ((v[:, :, 0] + v[:, :, 1]) * inv)[:37].reshape(37, cc * 32)[:, :72]
    err = (back - x.double()).abs()
    assert bool((err <= torch.maximum(x.double().abs() * 2.0 ** -22, torch.tensor(2.0 ** -24 * inv, dtype=torch.float64))).all())
    big = x.abs() >= bound * 2.0 ** -10
    assert float((err[big] / x.double().abs()[big]).max()) <= 2.0 ** -22
    assert float(v[37].abs().max()) == 0.0 and float(v[:37, 2, :, 8:].abs().max()) == 0.0
    # an all-zero tensor: scale 1, every part zero
    z = ops.split3_rows(torch.zeros(5, 32, device="cuda"), 32, 5, 32)
    assert float(z[-16:].view(torch.float32)[0]) == 1.0 and int(z[:-16].view(torch.int16).abs().max()) == 0


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("case", CASES)
def test_f16x3_forward_vs_f64_every_tile(case, tile):
    from dass_hip import ops
    from dass_hip._lib import lib

    lib.dass_x3_force_tile(tile)
    n, c, h, wd, k, ks, stride, pad, dil = case
    x, w = _inputs(case)
    ref = F.conv2d(x.double(), w.double(), None, stride, pad, dil).permute(0, 2, 3, 1)
    y = _fwd(ops, "f16x3", case, x, w)
    assert _rel(y, ref) <= 2e-6, (case, tile, _rel(y, ref))
    assert (y.double().cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("kind", ["gauss", "relu", "tiny", "large", "heavy", "outlier"])
@pytest.mark.parametrize("case", [CASES[0], CASES[3], CASES[4], CASES[6]])
def test_f16x3_error_is_at_the_f32_level(case, kind):
    """rel-L2 error against the f64 conv: the two-part engine within 2x of the plain f32 MFMA (which rounds every partial sum
    to 24 bits) on every input class, and never worse than 1.5x the six-product bf16 engine + 1e-7"""
    from dass_hip import ops

    n, c, h, wd, k, ks, stride, pad, dil = case
    x, w = _inputs(case, kind)
    ref = F.conv2d(x.double(), w.double(), None, stride, pad, dil).permute(0, 2, 3, 1)
    errs = {e: _rel(_fwd(ops, e, case, x, w), ref) for e in ("f16x3", "bf16x6", "f32")}
    print(case, kind, {e: "%.2e" % v for e, v in errs.items()})
    # a tensor dominated by one outlier / lognormal tails keeps its bulk many binades below the bound, where the low part runs
    # into f16's subnormal spacing: the error there is absolute (2^-25 / scale), still far inside the 1e-3 contract
    slack = 2e-6 if kind in ("heavy", "outlier") else 0.0
    assert errs["f16x3"] <= max(2.0 * errs["f32"] + 1e-7, slack), errs
    assert errs["f16x3"] <= max(1.5 * errs["bf16x6"] + 1e-7, slack), errs


@pytest.mark.parametrize("case", [CASES[0], CASES[2], CASES[4], CASES[6], CASES[7]])
def test_f16x3_gradients_vs_f64(case):
    """input gradient (the same kernel over dy and the flipped operand, phase-decomposed for stride 2) and weight gradient
    (dass_conv2d_wgrad_x3 on two-part rows) with gradient-sized dy (1e-6)"""
    from dass_hip import ops
    from dass_hip._lib import check, lib

    n, c, h, wd, k, ks, stride, pad, dil = case
    x, w = _inputs(case)
    oh, ow = ops.conv_out_size(h, ks, stride, pad, dil), ops.conv_out_size(wd, ks, stride, pad, dil)
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(n, k, oh, ow, generator=g) * 1e-6
    xd, wd64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    F.conv2d(xd, wd64, None, stride, pad, dil).backward(dy.double())
    dyr, xr = _rows(dy), _rows(x)
    dy3 = ops.split3_rows(dyr, k, n * oh * ow, k)
    wt3 = ops.prepare_conv_weight(w.permute(0, 2, 3, 1).contiguous().cuda(), mode=1, x3=True)
    dx = torch.full((n, h, wd, c), float("nan"), device="cuda")
    pad_t = dil * (ks - 1) - pad
    ops.conv_x3_launch(dy3, wt3, dx, c, (n, oh, ow, k, h, wd, c, ks, ks, 1, pad_t, dil), ustride=stride)
    assert _rel(dx, xd.grad.permute(0, 2, 3, 1)) <= 3e-6
    x3 = ops.split3_rows(xr, c, n * h * wd, c)
    dw = torch.empty((k, ks, ks, c), device="cuda")
    check(lib.dass_conv2d_wgrad_x3(ops._p(x3), ops._p(dy3), ops._p(dw), n, h, wd, c, oh, ow, k, ks, ks, stride, pad, dil, 1, ops._stream()),
          "wgrad_x3")
    assert _rel(dw, wd64.grad.permute(0, 2, 3, 1)) <= 3e-6


@pytest.mark.parametrize("backbone,seed", [("resnet", 31), ("mobilenet", 41)])
def test_train_step_f16x3_vs_oracle_and_bf16x6(backbone, seed):
    """one train-mode step (batch statistics): loss against the f64 oracle to 1e-5; the gradients of the f16x3 AND the bf16x6 step
    against the f64 oracle under each run's own gates, within the multiples of stock f32 PyTorch (same gates) that
    tests/gate_replay.py:assert_gated_step states -- no floors; gate flips against the f64 forward counted and bounded"""
    from dass_hip import ops
    from gate_replay import ENGINE_MULT, assert_gated_step, gated_step_report
    from models.deeplab import DeepLab
    from oracle import deeplab_cpu as O
    from oracle import selection_cpu as S
    from utils.loss import SegmentationLosses

    ncls, n, hw = 19, 4, 65
    om = O.ODeepLab(backbone, 16, ncls)
    O.fill_state_dict(om, seed=seed, randomize_bn_stats=False)
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=520)
    m1, m2 = O.dropout_masks(n, 1, seed=23)
    med = {}
    for engine in ("f16x3", "bf16x6"):
        ops.set_f32_mma(engine)
        pm = DeepLab(backbone=backbone, output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False)
        pm.load_state_dict(om.state_dict())
        pm = pm.cuda().train()
        rep = gated_step_report(ops, O, S, pm, om.state_dict(), backbone, ncls, x, lab, (m1[0], m2[0]), SegmentationLosses(cuda=True).build_loss("ce"))
        assert abs(rep["loss"] - rep["loss64"]) <= 1e-5 * abs(rep["loss64"]), (engine, rep["loss"], rep["loss64"])
        # (MobileNet: every block's expand / depthwise BN sits between the ASPP's batch-4 image-pool BN and nothing -- only the decoder
        #  and the ASPP conv branches are "downstream" in the sense of assert_gated_step)
        assert_gated_step(rep, "%s %s" % (backbone, engine), mult=ENGINE_MULT[engine])
        med[engine] = float(np.median(list(rep["err_inj"].values())))
    # (the six-product engine multiplies EXACT operands and is several times closer to f64 than any f32-input arithmetic; the
    # two-part engine rounds operands to 23 bits and lands between it and stock f32)
    print("   f16x3 / bf16x6 median ratio %.1f" % (med["f16x3"] / max(med["bf16x6"], 1e-12)))


def test_mc_dropout_votes_f16x3_vs_oracle():
    from dass_hip import ops
    from models.deeplab import DeepLab
    from oracle import deeplab_cpu as O
    from oracle import selection_cpu as S

    ncls, n, hw, T = 19, 3, 65, 4
    om = O.ODeepLab("resnet", 16, ncls)
    O.fill_state_dict(om, seed=17)
    om.eval()
    pm = DeepLab(backbone="resnet", output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False)
    pm.load_state_dict(om.state_dict())
    pm = pm.cuda().eval()
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=77)
    m1, m2 = O.dropout_masks(n, T, seed=9)
    with torch.no_grad():
        want = om(x)
        got = pm(x.cuda()).float().cpu()
    assert (got - want).abs().max().item() <= 1e-3
    votes = pm.mc_dropout_votes(x.cuda(), T, masks=(m1, m2))
    ref = S.mc_votes(om, x, (m1, m2))
    assert float((votes.cpu().long() != ref).float().mean()) <= 1e-4


@pytest.mark.parametrize("engine", ["f16x3", "bf16x6"])
def test_grouped_weight_gradients_vs_f64(engine):
    """dass_conv2d_wgrad_x3_group: five layers of both tile classes (one of them long enough to be cut into pixel ranges) in ONE
    call, each against the f64 weight gradient"""
    import ctypes

    from dass_hip import ops
    from dass_hip._lib import check, lib

    ops.set_f32_mma(engine)
    cases = [CASES[0], CASES[2], CASES[4], CASES[6], CASES[7], (2, 64, 129, 129, 64, 3, 1, 1, 1)]
    items = np.zeros((len(cases), 16), dtype=np.int64)
    keep, refs, outs = [], [], []
    for i, case in enumerate(cases):
        n, c, h, wd, k, ks, stride, pad, dil = case
        x, w = _inputs(case)
        oh, ow = ops.conv_out_size(h, ks, stride, pad, dil), ops.conv_out_size(wd, ks, stride, pad, dil)
        dy = torch.randn(n, k, oh, ow, generator=torch.Generator().manual_seed(5 + i)) * 1e-5
        wd64 = w.double().requires_grad_(True)
        F.conv2d(x.double(), wd64, None, stride, pad, dil).backward(dy.double())
        refs.append(wd64.grad.permute(0, 2, 3, 1))
        x3 = ops.split3_rows(_rows(x), c, n * h * wd, c)
        dy3 = ops.split3_rows(_rows(dy), k, n * oh * ow, k)
        dw = torch.zeros((k, ks, ks, c), device="cuda")
        keep += [x3, dy3]
        outs.append(dw)
        items[i, :3] = (x3.data_ptr(), dy3.data_ptr(), dw.data_ptr())
        items[i, 3:15] = (n, h, wd, c, oh, ow, k, ks, ks, stride, pad, dil)
    scratch = torch.empty((lib.dass_conv2d_wgrad_x3_group_scratch_bytes(len(cases)) + 128,), dtype=torch.uint8, device="cuda")
    check(lib.dass_conv2d_wgrad_x3_group(items.ctypes.data_as(ctypes.c_void_p), len(cases), ops._p(scratch), scratch.numel(), ops._stream()),
          "group")
    for case, dw, ref in zip(cases, outs, refs):
        assert _rel(dw, ref) <= 3e-6, (case, _rel(dw, ref))


def test_deferred_grouped_wgrad_equals_per_layer_launches():
    """a train step with the weight gradients deferred to the grouped launch at the end of backward == the same step with
    per-layer launches inside backward (f32 atomics in both: 4e-6), including gradient ACCUMULATION over two backward passes"""
    from dass_hip import ops
    from models.deeplab import DeepLab
    from oracle import deeplab_cpu as O
    from utils.loss import SegmentationLosses

    ncls, n, hw = 19, 2, 65
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=70)
    m1, m2 = O.dropout_masks(n, 1, seed=4)
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    torch.manual_seed(5)
    pm = DeepLab(backbone="resnet", output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False).cuda().train()
    res = {}
    keep = ops.deferred_wgrad()
    try:
        for mode in (True, False):
            ops.set_deferred_wgrad(mode)
            pm.zero_grad(set_to_none=True)
            for rep in range(2):   # the second pass accumulates into existing .grad tensors
                crit(pm(x.cuda(), dropout_masks=(m1[0].cuda(), m2[0].cuda())), lab.cuda()).backward()
            res[mode] = {k: p.grad.detach().double().cpu() for k, p in pm.named_parameters()}
    finally:
        ops.set_deferred_wgrad(keep)
    assert all(v is not None for v in res[True].values())
    worst = max(((res[True][k] - g).norm().item() / max(g.norm().item(), 1e-12), k) for k, g in res[False].items())
    assert worst[0] <= 1e-4, worst   # (BN running statistics moved between the two runs' passes: train-mode batch statistics do not depend on them)


def test_chunked_flush_of_deferred_wgrads_fires_every_hook_once():
    """ops.set_wgrad_chunk(n): the queue of deferred weight gradients is flushed every n layers (what dist.GradientAverager asks
    for with more than one process, so that the all-reduce of the first buckets runs under the rest of backward).  Same
    gradients as the single launch at the end, and every parameter's post-accumulate hook counts exactly one REAL call --
    calls made while `ops.wgrad_pending(p)` are autograd's empty-handed ones and are skipped, as the averager does."""
    from dass_hip import ops
    from models.deeplab import DeepLab
    from oracle import deeplab_cpu as O
    from utils.loss import SegmentationLosses

    ncls, n, hw = 19, 2, 65
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=90)
    m1, m2 = O.dropout_masks(n, 1, seed=6)
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    torch.manual_seed(9)
    pm = DeepLab(backbone="resnet", output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False).cuda().train()
    calls = {}
    handles = []
    for name, p in pm.named_parameters():
        def hook(q, name=name):
            if ops.wgrad_pending(q):
                return
            assert q.grad is not None, name
            calls[name] = calls.get(name, 0) + 1
        handles.append(p.register_post_accumulate_grad_hook(hook))
    res = {}
    keep = (ops._wg["chunk"], ops._wg["side"])
    try:
        for chunk, side in ((0, False), (7, False), (7, True), (16, True)):   # (16, True) is the default
            ops.set_wgrad_chunk(chunk, side)
            for rep in range(2):   # twice: the second pass ACCUMULATES into .grad while side launches may still be running
                if rep == 0:
                    pm.zero_grad(set_to_none=True)
                calls.clear()
                crit(pm(x.cuda(), dropout_masks=(m1[0].cuda(), m2[0].cuda())), lab.cuda()).backward()
                wrong = {k: v for k, v in calls.items() if v != 1}
                missing = [k for k, _ in pm.named_parameters() if k not in calls]
                assert not wrong and not missing, (chunk, side, rep, wrong, missing[:5])
            res[(chunk, side)] = {k: p.grad.detach().double().cpu() for k, p in pm.named_parameters()}
    finally:
        ops.set_wgrad_chunk(*keep)
        for h in handles:
            h.remove()
    for key in res:
        worst = max(((res[key][k] - g).norm().item() / max(g.norm().item(), 1e-12), k) for k, g in res[(0, False)].items())
        assert worst[0] <= 1e-4, (key, worst)   # (BN running statistics move between the passes: batch statistics do not depend on them)


def test_deferred_wgrads_survive_a_backward_pass_that_raised():
    """a backward pass that dies half-way never runs autograd's final callback: its queued weight gradients must not leak into the
    next pass, and the next pass must arm its own callback (every parameter gets its gradient)"""
    from dass_hip import ops
    from models.deeplab import DeepLab
    from oracle import deeplab_cpu as O
    from utils.loss import SegmentationLosses

    ncls, n, hw = 19, 2, 65
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=95)
    m1, m2 = O.dropout_masks(n, 1, seed=8)
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    torch.manual_seed(3)
    pm = DeepLab(backbone="resnet", output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False).cuda().train()

    def run(poison):
        pm.zero_grad(set_to_none=True)
        out = pm(x.cuda(), dropout_masks=(m1[0].cuda(), m2[0].cuda()))
        if poison:
            # an op in the middle of the graph whose backward raises (after the decoder / ASPP layers have queued their gradients)
            def boom(g):
                raise RuntimeError("injected failure")
            handle = pm.backbone.layer4[0].bn1.weight.register_hook(boom)   # fires in the middle of the pass
            try:
                with pytest.raises(RuntimeError, match="injected failure"):
                    crit(out, lab.cuda()).backward()
            finally:
                handle.remove()
            return None
        crit(out, lab.cuda()).backward()
        return {k: (None if p.grad is None else p.grad.detach().double().cpu()) for k, p in pm.named_parameters()}

    ref = run(False)
    run(True)
    got = run(False)
    assert all(v is not None for v in got.values())
    worst = max(((got[k] - g).norm().item() / max(g.norm().item(), 1e-12), k) for k, g in ref.items())
    assert worst[0] <= 1e-4, worst
