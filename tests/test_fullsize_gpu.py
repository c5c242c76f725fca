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
