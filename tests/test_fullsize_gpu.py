"""BASELINE.json's full sizes (DeepLab-R101 os16, 19 classes, 513x513; decoder 3x3 304->256 @129x129 batch 8): the CPU
oracle would take minutes there, so parity is carried by size-independent properties of the path:

  * exact linearity of the convs in their activation operand: scaling an operand by 2 scales forward, input gradient
    and weight gradient by exactly 2, bit for bit, in every engine (a power of two commutes with every rounding on the
    path, the bf16 three-way split included) -- any tile/tap/edge indexing slip at the big shapes breaks it;
  * two independent conv engines (three-way bf16 split on the bf16 MFMA pipe vs the plain f32 MFMA) agree on full-size
    logits to the parity tolerance (1e-3) and on every argmax outside near-ties;
  * MC-dropout votes: the hoisted T-pass path equals T full forwards with the same masks, the vote histogram sums to T,
    the vote entropy (log2, mc_dropout.py:46-48) lies in [0, log2 min(T, C)], and a batch scored in two shards gives the same votes / scores as in
    one (the property the multi-GPU pool sharding rests on).
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_mma_mode():
    from dass_hip import ops

    mode = ops.f32_mma()
    ops.set_compute_dtype(torch.float32)
    yield
    ops.set_f32_mma(mode)


def _r101(seed=5):
    from oracle import deeplab_cpu as O
    from models.deeplab import DeepLab

    om = O.ODeepLab("resnet101", 16, 19)
    O.fill_state_dict(om, seed=seed)
    pm = DeepLab(backbone="resnet101", output_stride=16, num_classes=19, sync_bn=False, freeze_bn=False, pretrained=False)
    pm.load_state_dict(om.state_dict())
    return pm.cuda().eval(), O


@pytest.mark.parametrize("engine", ["bf16x6", "f32", "bf16x3"])
def test_conv_exact_linearity_at_the_roofline_shape(engine):
    from dass_hip import ops
    from dass_hip._lib import check, lib

    ops.set_f32_mma(engine)
    n, h, c, k = 8, 129, 304, 256
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn((n, h, h, c), device="cuda", generator=g)
    wt = torch.randn((k, 3, 3, c), device="cuda", generator=g) * 0.02
    dy = torch.randn((n, h, h, k), device="cuda", generator=g) * 1e-3
    dims = (n, h, h, c, h, h, k, 3, 3, 1, 1, 1)
    wop = ops.prepare_conv_weight(wt)

    def fwd(inp):
        y = torch.empty((n, h, h, k), device="cuda")
        ops.conv_launch(inp, c, wop, y, k, dims)
        return y

    def wgrad(inp, grad):
        dw = torch.empty((k, 3, 3, c), device="cuda")
        check(lib.dass_conv2d_wgrad(ops._p(inp), c, ops._p(grad), k, ops._p(dw), n, h, h, c, h, h, k, 3, 3, 1, 1, 1, ops._cdt(grad),
                                    ops._stream()), "wgrad")
        return dw

    y1, y2 = fwd(x), fwd(x * 2.0)
    assert torch.isfinite(y1).all() and y1.abs().max() > 0
    assert torch.equal(y2, y1 * 2.0), "forward conv is not exactly linear in x"
    # dgrad = the same kernel over the flipped [C][3][3][K] operand, fed with dy
    wt_t = wt.permute(3, 1, 2, 0).flip(1, 2).contiguous()
    wop_t = ops.prepare_conv_weight(wt_t)
    dims_t = (n, h, h, k, h, h, c, 3, 3, 1, 1, 1)

    def dgrad(grad):
        dx = torch.empty((n, h, h, c), device="cuda")
        ops.conv_launch(grad, k, wop_t, dx, c, dims_t)
        return dx

    d1, d2 = dgrad(dy), dgrad(dy * 2.0)
    assert torch.equal(d2, d1 * 2.0), "dgrad is not exactly linear in dy"
    w1, w2, w3 = wgrad(x, dy), wgrad(x, dy * 2.0), wgrad(x * 2.0, dy)
    # atomics change the summation order between launches: linearity is exact per partial, the sum agrees to rounding
    tol = 4e-6 * w1.abs().max().item()
    assert (w2 - 2.0 * w1).abs().max().item() <= tol and (w3 - 2.0 * w1).abs().max().item() <= tol
    # edge rows / columns of the image (padding taps) against an independent torch conv on a corner crop
    ref = torch.nn.functional.conv2d(x[:1, :6, :6].permute(0, 3, 1, 2).double().cpu(), wt.permute(0, 3, 1, 2).double().cpu(), padding=1)
    got = y1[:1, :5, :5].permute(0, 3, 1, 2).double().cpu()
    assert (got - ref[:, :, :5, :5]).abs().max().item() <= (2e-5 if engine == "bf16x3" else 5e-6) * ref.abs().max().item()


def test_engines_agree_on_full_size_logits():
    from dass_hip import ops

    pm, O = _r101()
    x, _ = O.synthetic_batch(2, 513, 513, 19, first_index=900)
    outs = {}
    with torch.no_grad():
        for engine in ("bf16x6", "f32"):
            ops.set_f32_mma(engine)
            outs[engine] = pm(x.cuda()).float()
    a, b = outs["bf16x6"], outs["f32"]
    assert a.shape == (2, 19, 513, 513)
    err = (a - b).abs().max().item()
    top = b.topk(2, dim=1)[0]
    safe = (top[:, 0] - top[:, 1]) > 1e-3
    flips = int((a.argmax(1) != b.argmax(1)).sum())
    print("full-size R101 513^2: engines differ by %.2e (logit scale %.1f), argmax flips %d, near-ties %d"
          % (err, b.abs().max().item(), flips, int((~safe).sum())))
    assert err <= 1e-3
    assert torch.equal(a.argmax(1)[safe], b.argmax(1)[safe])


def test_mc_dropout_properties_full_size():
    from dass_hip import ops

    pm, O = _r101(seed=6)
    n, T, ncls = 4, 10, 19
    x, lab = O.synthetic_batch(n, 513, 513, ncls, first_index=950)
    m1, m2 = O.dropout_masks(n, T, seed=9)  # deterministic Bernoulli multipliers
    xd = x.cuda()
    votes = pm.mc_dropout_votes(xd, T, masks=(m1, m2))
    assert votes.shape == (n, T, 513, 513) and votes.dtype == torch.uint8 and int(votes.max()) < ncls
    # (1) deterministic, and equal to T full forwards with the same masks (the reference's way)
    assert torch.equal(votes, pm.mc_dropout_votes(xd, T, masks=(m1, m2)))
    with torch.no_grad():
        for t in (0, T - 1):
            full = pm(xd, dropout_masks=(m1[t].cuda(), m2[t].cuda()))
            top = full.topk(2, dim=1)[0]
            safe = (top[:, 0] - top[:, 1]) > 1e-3
            assert torch.equal(full.argmax(1)[safe].to(torch.uint8), votes[:, t][safe])
    # (2) histogram sums to T; entropy within its bounds
    hist = torch.stack([(votes == c).sum(1) for c in range(ncls)], 1)
    assert int(hist.sum(1).min()) == T and int(hist.sum(1).max()) == T
    emap, means = ops.vote_entropy(votes, lab.cuda(), ncls)
    assert emap.min().item() >= -1e-6 and emap.max().item() <= math.log2(min(T, ncls)) + 1e-5
    p = hist.float() / T
    ref_e = -(p * torch.log2(p + 1e-12)).sum(1)  # the reference's formula
    valid = (lab.cuda() >= 0) & (lab.cuda() < ncls)
    assert (emap - ref_e * valid).abs().max().item() <= 2e-5
    # (3) sharding invariance: two shards of 2 images == one batch of 4 (eval-mode BN: images are independent)
    va = pm.mc_dropout_votes(xd[:2], T, masks=(m1[:, :2], m2[:, :2]))
    vb = pm.mc_dropout_votes(xd[2:], T, masks=(m1[:, 2:], m2[:, 2:]))
    sharded = torch.cat((va, vb), 0)
    diff = int((sharded != votes).sum())
    print("MC-dropout full size: sharded-vs-whole vote differences %d of %d" % (diff, votes.numel()))
    assert diff <= votes.numel() * 1e-5  # tile decomposition differs with the batch: only exact near-ties may move
    _, means_s = ops.vote_entropy(sharded, lab.cuda(), ncls)
    assert (means_s - means).abs().max().item() <= 1e-4


# ----------------------------------------------------------------------------------------------------------------------
# Independent value checks at full size: >= 4096 randomly chosen output pixels -- every image, all four
# borders, the corners, the last M-tile and (through all K columns) the last N-tile -- of forward, input gradient and a
# weight-gradient slice, against f64 dot products computed OUTSIDE the kernels (torch f64 over gathered 3x3 patches).
def _sample_pixels(n, h, w, count, seed):
    g = torch.Generator().manual_seed(seed)
    idx = torch.stack((torch.randint(0, n, (count,), generator=g), torch.randint(0, h, (count,), generator=g),
                       torch.randint(0, w, (count,), generator=g)), 1)
    edge = []
    for img in range(n):  # corners and a point on every border of every image; the very last pixel = last row of the last M-tile
        for (yy, xx) in ((0, 0), (0, w - 1), (h - 1, 0), (h - 1, w - 1), (0, w // 2), (h - 1, w // 3), (h // 2, 0), (h // 3, w - 1)):
            edge.append((img, yy, xx))
    return torch.cat((idx, torch.tensor(edge)), 0)


def _patches_f64(x_nhwc, pix, ks, pad, dil):
    """x [N,H,W,C] (device) -> f64 [P, ks*ks*C]: the zero-padded receptive fields of the sampled pixels (stride 1)"""
    n, h, w, c = x_nhwc.shape
    xp = torch.zeros((n, h + 2 * pad, w + 2 * pad, c), dtype=torch.float64, device=x_nhwc.device)
    xp[:, pad:pad + h, pad:pad + w] = x_nhwc.double()
    cols = []
    for r in range(ks):
        for s in range(ks):
            cols.append(xp[pix[:, 0], pix[:, 1] + r * dil, pix[:, 2] + s * dil])
    return torch.cat(cols, 1)


@pytest.mark.parametrize("shape", [(8, 129, 304, 256), (8, 193, 304, 256), (8, 33, 256, 256)])
@pytest.mark.parametrize("engine", ["bf16x6", "bf16x6+x3", "f32"])
def test_fullsize_conv_values_vs_f64_dot_products(shape, engine):
    from dass_hip import ops
    from dass_hip._lib import check, lib

    n, hw, c, k = shape
    x3_engine = engine.endswith("x3")
    ops.set_f32_mma(engine.split("+")[0])
    g = torch.Generator(device="cuda").manual_seed(n * hw + c)
    x = torch.randn((n, hw, hw, c), device="cuda", generator=g)
    wt = torch.randn((k, 3, 3, c), device="cuda", generator=g) * (2.0 / (9 * c)) ** 0.5
    dy = torch.randn((n, hw, hw, k), device="cuda", generator=g)
    dims = (n, hw, hw, c, hw, hw, k, 3, 3, 1, 1, 1)
    pix = _sample_pixels(n, hw, hw, 4096, seed=hw).cuda()
    w64 = wt.double()

    # ---- forward
    y = torch.full((n, hw, hw, k), float("nan"), device="cuda")
    if x3_engine:
        x3 = ops.split3_rows(x, c, n * hw * hw, c)
        ops.conv_x3_launch(x3, ops.prepare_conv_weight(wt), y, k, dims)
    else:
        ops.conv_launch(x, c, ops.prepare_conv_weight(wt), y, k, dims)
    ref = _patches_f64(x, pix, 3, 1, 1) @ w64.reshape(k, -1).t()
    got = y[pix[:, 0], pix[:, 1], pix[:, 2]].double()
    scale = ref.abs().max().item()
    assert torch.isfinite(y).all()
    assert (got - ref).abs().max().item() <= 2e-5 * scale, ("fwd", (got - ref).abs().max().item(), scale)

    # ---- input gradient: dx[n,y,x,:] = sum over taps of dy[n, y+1-r, x+1-s, :] @ W[:, r, s, :]
    dx = torch.full((n, hw, hw, c), float("nan"), device="cuda")
    w_t = ops.prepare_conv_weight(wt.permute(3, 1, 2, 0).flip(1, 2).contiguous())  # [C][3][3][K], taps flipped
    ddims = (n, hw, hw, k, hw, hw, c, 3, 3, 1, 1, 1)
    if x3_engine:
        dy3 = ops.split3_rows(dy, k, n * hw * hw, k)
        ops.conv_x3_launch(dy3, w_t, dx, c, ddims)
    else:
        ops.conv_launch(dy, k, w_t, dx, c, ddims)
    wflip = w64.flip(1, 2).permute(1, 2, 0, 3).reshape(9 * k, c)   # [(r', s', k), c] with r' = 2 - r
    refd = _patches_f64(dy, pix, 3, 1, 1) @ wflip
    gotd = dx[pix[:, 0], pix[:, 1], pix[:, 2]].double()
    assert torch.isfinite(dx).all()
    assert (gotd - refd).abs().max().item() <= 2e-5 * refd.abs().max().item(), "dgrad"

    # ---- weight gradient, 8 output channels spread over the K tiles (including the last one), all taps and input channels
    ksel = torch.tensor([0, 31, 64, 127, 128, 200, k - 2, k - 1], device="cuda")
    dw = torch.full((k, 3, 3, c), float("nan"), device="cuda")
    if x3_engine:
        x3, dy3 = ops.split3_rows(x, c, n * hw * hw, c), ops.split3_rows(dy, k, n * hw * hw, k)  # held until the launch is enqueued
        check(lib.dass_conv2d_wgrad_x3(ops._p(x3), ops._p(dy3), ops._p(dw), n, hw, hw, c, hw, hw, k, 3, 3, 1, 1, 1, 1, ops._stream()),
              "dass_conv2d_wgrad_x3")
    else:
        check(lib.dass_conv2d_wgrad(ops._p(x), c, ops._p(dy), k, ops._p(dw), n, hw, hw, c, hw, hw, k, 3, 3, 1, 1, 1,
                                    ops._cdt(x), ops._stream()), "dass_conv2d_wgrad")
    xp = torch.zeros((n, hw + 2, hw + 2, c), dtype=torch.float64, device="cuda")
    xp[:, 1:hw + 1, 1:hw + 1] = x.double()
    dys = dy[..., ksel].double().reshape(-1, ksel.numel())
    refw = torch.stack([torch.stack([dys.t() @ xp[:, r:r + hw, s:s + hw].reshape(-1, c) for s in range(3)], 1) for r in range(3)], 1)
    gotw = dw[ksel].double()
    assert torch.isfinite(dw).all()
    assert (gotw - refw).abs().max().item() <= 1e-5 * refw.abs().max().item(), "wgrad"


@pytest.mark.parametrize("shape", [(8, 129, 304, 256, 3), (8, 193, 304, 256, 3), (8, 33, 256, 256, 3), (8, 129, 64, 256, 1), (8, 193, 256, 64, 1)])
@pytest.mark.parametrize("tile", [0, 14, 11])
def test_fullsize_f16x3_values_vs_f64_dot_products(shape, tile):
    """The DEFAULT engine at the headline shapes: the two-part pre-split kernels -- the dispatcher's choice (tile 0),
    the whole-tile 64 x 64 specialisation forced (14: magic-multiplier pixel decomposition, 32-bit epilogue offsets, folded tap
    barrier, at M = 133 128 and M = 297 992 = 18 628 workgroups) and whole 256 x 128 tiles (11) -- forward, input gradient and the
    GROUPED weight-gradient launch the train step uses, against f64 dot products over gathered patches computed outside the
    kernels; 4096 random pixels + every corner / border of every image (last M- and N-tiles included)."""
    import ctypes

    import numpy as np
    from dass_hip import ops
    from dass_hip._lib import check, lib

    n, hw, c, k, ks = shape
    pad = ks // 2
    ops.set_f32_mma("f16x3")
    g = torch.Generator(device="cuda").manual_seed(n * hw + c + ks)
    x = torch.randn((n, hw, hw, c), device="cuda", generator=g)
    wt = torch.randn((k, ks, ks, c), device="cuda", generator=g) * (2.0 / (ks * ks * c)) ** 0.5
    dy = torch.randn((n, hw, hw, k), device="cuda", generator=g) * 1e-3   # gradient-sized
    dims = (n, hw, hw, c, hw, hw, k, ks, ks, 1, pad, 1)
    pix = _sample_pixels(n, hw, hw, 4096, seed=hw + ks).cuda()
    w64 = wt.double()
    try:
        lib.dass_x3_force_tile(tile)
        # ---- forward
        y = torch.full((n, hw, hw, k), float("nan"), device="cuda")
        x3 = ops.split3_rows(x, c, n * hw * hw, c)
        ops.conv_x3_launch(x3, ops.prepare_conv_weight(wt, x3=True), y, k, dims)
        pick_f = lib.dass_x3_last_pick()
        ref = _patches_f64(x, pix, ks, pad, 1) @ w64.reshape(k, -1).t()
        got = y[pix[:, 0], pix[:, 1], pix[:, 2]].double()
        assert torch.isfinite(y).all()
        assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item(), ("fwd", (got - ref).abs().max().item(), ref.abs().max().item())
        if tile == 14:
            assert pick_f >> 16 == 64 and (pick_f >> 4) & 0xfff == 64 and pick_f & 2, hex(pick_f)   # the whole-tile kernel really ran
        # ---- input gradient
        dx = torch.full((n, hw, hw, c), float("nan"), device="cuda")
        w_t = ops.prepare_conv_weight(wt.permute(3, 1, 2, 0).flip(1, 2).contiguous(), x3=True)
        dy3 = ops.split3_rows(dy, k, n * hw * hw, k)
        ops.conv_x3_launch(dy3, w_t, dx, c, (n, hw, hw, k, hw, hw, c, ks, ks, 1, pad, 1))
        wflip = w64.flip(1, 2).permute(1, 2, 0, 3).reshape(ks * ks * k, c)
        refd = _patches_f64(dy, pix, ks, pad, 1) @ wflip
        gotd = dx[pix[:, 0], pix[:, 1], pix[:, 2]].double()
        assert torch.isfinite(dx).all()
        assert (gotd - refd).abs().max().item() <= 2e-5 * refd.abs().max().item(), "dgrad"
    finally:
        lib.dass_x3_force_tile(0)
    if tile != 0:
        return  # (the weight-gradient launch has its own tile classes: once per shape)
    # ---- grouped weight gradient (one problem), 8 output channels over the K tiles, all taps and input channels
    ksel = torch.tensor(sorted({0, 31, min(64, k - 1), k // 2 - 1, k // 2, k - 3, k - 2, k - 1}), device="cuda")
    dw = torch.zeros((k, ks, ks, c), device="cuda")
    items = np.zeros((1, 16), dtype=np.int64)
    items[0, :3] = (x3.data_ptr(), dy3.data_ptr(), dw.data_ptr())
    items[0, 3:15] = dims
    scratch = torch.empty((lib.dass_conv2d_wgrad_x3_group_scratch_bytes(1) + 128,), dtype=torch.uint8, device="cuda")
    check(lib.dass_conv2d_wgrad_x3_group(items.ctypes.data_as(ctypes.c_void_p), 1, ops._p(scratch), scratch.numel(), ops._stream()), "group")
    xp = torch.zeros((n, hw + 2 * pad, hw + 2 * pad, c), dtype=torch.float64, device="cuda")
    xp[:, pad:hw + pad, pad:hw + pad] = x.double()
    dys = dy[..., ksel].double().reshape(-1, ksel.numel())
    refw = torch.stack([torch.stack([dys.t() @ xp[:, r:r + hw, s2:s2 + hw].reshape(-1, c) for s2 in range(ks)], 1) for r in range(ks)], 1)
    gotw = dw[ksel].double()
    assert torch.isfinite(dw).all()
    assert (gotw - refw).abs().max().item() <= 1e-5 * refw.abs().max().item(), ("wgrad", (gotw - refw).abs().max().item(), refw.abs().max().item())


def test_config_b_r101_769_logits_vs_oracle():
    """BASELINE config B (R101 os16, 769 x 769) against the CPU oracle DIRECTLY (one image costs seconds on the box's
    host cores, no need to argue it through properties): eval logits of one image within the contract's ABSOLUTE 1e-3 (measured
    1.7e-4 on a logit scale of 60.6), argmax identical where the top-2 margin exceeds 1e-3.  Default engine (f16x3)."""
    pm, O = _r101(seed=7)
    om = O.ODeepLab("resnet101", 16, 19)
    O.fill_state_dict(om, seed=7)
    om.eval()
    x, _ = O.synthetic_batch(1, 769, 769, 19, first_index=31)
    torch.set_num_threads(min(16, torch.get_num_threads() or 16))
    with torch.no_grad():
        ref = om(x).float()
        got = pm(x.cuda()).float().cpu()
    assert got.shape == ref.shape == (1, 19, 769, 769)
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item()
    top = ref.topk(2, dim=1)[0]
    margin = 1e-3
    safe = (top[:, 0] - top[:, 1]) > margin
    flips = int((got.argmax(1) != ref.argmax(1)).sum())
    print("config B 769^2 vs oracle: max |dlogit| %.2e on a logit scale of %.1f, argmax flips %d, near-ties %d" % (err, scale, flips, int((~safe).sum())))
    assert err <= margin, (err, scale)
    assert torch.equal(got.argmax(1)[safe], ref.argmax(1)[safe])


def test_config_b_r101_769_forward_and_mc_dropout():
    """BASELINE config B's size (R101 os16, 769 x 769): forward through both parity engine paths (classic kernels under
    autograd bookkeeping off = the pre-split inference engine, and with it switched off) agree to the parity tolerance with
    identical argmax outside near-ties; MC-dropout: hoisted T passes == T full forwards, histogram sums to T, entropy in
    range.  193 x 193 decoder maps, 49 x 49 ASPP maps: tile edges no 513 test touches."""
    from dass_hip import ops

    pm, O = _r101(seed=7)
    n, hw, T = 2, 769, 3
    x, lab = O.synthetic_batch(n, hw, hw, 19, first_index=30)
    xd = x.cuda()
    outs = {}
    keep = ops.x3_mode()
    try:
        for mode in ("infer", "off"):
            ops.set_x3_pipeline(mode)
            with torch.no_grad():
                outs[mode] = pm(xd).float()
    finally:
        ops.set_x3_pipeline(keep)
    a, b = outs["infer"], outs["off"]
    assert a.shape == (n, 19, hw, hw) and torch.isfinite(a).all()
    scale = a.abs().max().item()
    assert (a - b).abs().max().item() <= 1e-3 * max(1.0, scale / 50), ((a - b).abs().max().item(), scale)
    top = a.topk(2, dim=1)[0]
    safe = (top[:, 0] - top[:, 1]) > 1e-3 * max(1.0, scale / 50)
    assert torch.equal(a.argmax(1)[safe], b.argmax(1)[safe])
    m1, m2 = O.dropout_masks(n, T, seed=8)
    votes = pm.mc_dropout_votes(xd, T, masks=(m1, m2))
    with torch.no_grad():
        full = torch.stack([pm(xd, dropout_masks=(m1[t].cuda(), m2[t].cuda())).argmax(1) for t in range(T)], 1)
    assert votes.shape == (n, T, hw, hw)
    assert (votes.long() != full).float().mean().item() <= 1e-4   # near-tie pixels only
    emap, means = ops.vote_entropy(votes, lab.cuda(), 19)
    assert float(emap.min()) >= 0.0 and float(emap.max()) <= math.log2(min(T, 19)) + 1e-5
    assert float(emap[:, : hw // 10].abs().max()) == 0.0   # ignore rows (label 255) score 0


def test_config_c_mobilenet_voc_batch16_train_steps():
    """BASELINE config C: DeepLab-MobileNetV2, 21 classes, 513 x 513, batch 16 -- three SGD steps on one fixed batch through
    the product surface; the first-step loss (train-mode BN) equals the CPU oracle's, every parameter receives a finite
    gradient, and the loss goes down."""
    from dass_hip import ops
    from dass_hip.optim import SGD
    from models.deeplab import DeepLab
    from oracle import deeplab_cpu as O
    from oracle import selection_cpu as S
    from utils.loss import SegmentationLosses

    ncls, n, hw = 21, 16, 513
    om = O.ODeepLab("mobilenet", 16, ncls)
    O.fill_state_dict(om, seed=41, randomize_bn_stats=False)
    pm = DeepLab(backbone="mobilenet", output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False)
    pm.load_state_dict(om.state_dict())
    pm = pm.cuda().train()
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=60)
    m1, m2 = O.dropout_masks(n, 3, seed=42)
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    opt = SGD([{"params": pm.get_1x_lr_params(), "lr": 0.007}, {"params": pm.get_10x_lr_params(), "lr": 0.07}], momentum=0.9, weight_decay=5e-4)
    om.train()
    torch.set_num_threads(min(16, torch.get_num_threads() or 16))
    with torch.no_grad():
        ref_loss = float(S.ce_loss(om(x, (m1[0], m2[0])), lab))
    losses = []
    xd, ld = x.cuda(), lab.cuda()
    for step in range(4):
        opt.zero_grad(set_to_none=True)
        loss = crit(pm(xd, dropout_masks=(m1[0].cuda(), m2[0].cuda())), ld)   # one fixed objective: same batch, same masks
        loss.backward()
        if step == 0:
            missing = [k for k, p in pm.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
            assert not missing, missing[:5]
        opt.step()
        losses.append(loss.item())
    assert abs(losses[0] - ref_loss) <= 2e-4 * abs(ref_loss), (losses[0], ref_loss)
    assert losses[3] < losses[0], losses


def test_config_c_full_size_gradients_under_the_hip_gates():
    """BASELINE config C at its size (MobileNetV2, 21 classes, 513 x 513; batch 4 keeps the f64 oracle passes at seconds): every parameter gradient
    of one train-mode step against the f64 oracle UNDER THE HIP FORWARD'S OWN GATES (ReLU6: both kinks), bounded by the f16x3 multiples of stock f32
    PyTorch under the same gates, gate flips against the f64 forward counted (tests/gate_replay.py) -- the strip depthwise kernels, the BN sums that
    ride in them and the expand / project 1x1 layers at 257^2 ... 33^2, none of which a 65^2 test reaches at these tile counts."""
    from dass_hip import ops
    from gate_replay import ENGINE_MULT, assert_gated_step, gated_step_report
    from models.deeplab import DeepLab
    from oracle import deeplab_cpu as O
    from oracle import selection_cpu as S
    from utils.loss import SegmentationLosses

    ncls, n, hw = 21, 4, 513
    assert ops.f32_mma() == "f16x3"
    om = O.ODeepLab("mobilenet", 16, ncls)
    O.fill_state_dict(om, seed=43, randomize_bn_stats=False)
    pm = DeepLab(backbone="mobilenet", output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False)
    pm.load_state_dict(om.state_dict())
    pm = pm.cuda().train()
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=77)
    m1, m2 = O.dropout_masks(n, 1, seed=44)
    rep = gated_step_report(ops, O, S, pm, om.state_dict(), "mobilenet", ncls, x, lab, (m1[0], m2[0]), SegmentationLosses(cuda=True).build_loss("ce"))
    print("config C train step 4 x 513^2: loss %.6f (f64 oracle %.6f)" % (rep["loss"], rep["loss64"]))
    assert abs(rep["loss"] - rep["loss64"]) <= 2e-4 * abs(rep["loss64"]), (rep["loss"], rep["loss64"])
    assert_gated_step(rep, "config C", mult=ENGINE_MULT["f16x3"])
