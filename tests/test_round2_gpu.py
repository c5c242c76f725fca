"""Round-2 regression tests of the host logic around the kernels: caches keyed on tensor versions must see the updates
the HIP optimizer / BN finalize kernels make through raw pointers; the image-level MC-dropout selector against the
oracle's stable top-k with tied scores."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup():
    from dass_hip import ops

    ops.set_compute_dtype(torch.float32)
    from oracle import deeplab_cpu as O
    from oracle import selection_cpu as S

    return ops, O, S


@pytest.fixture(autouse=True)
def _restore_modes():
    from dass_hip import ops

    mode, dt = ops.f32_mma(), ops.compute_dtype()
    yield
    ops.set_f32_mma(mode)
    ops.set_compute_dtype(dt)


@pytest.mark.parametrize("engine", ["f16x3", "bf16x6", "f32", "bf16"])
def test_hip_sgd_refreshes_weight_operands(engine):
    """dass_hip.optim.SGD writes parameters through raw pointers; the split / transposed / bf16 weight operands and the
    eval-BN vectors are cached on (data_ptr, _version).  Three train steps + an eval forward with the HIP optimizer
    must track the same model stepped by torch.optim.SGD (which bumps versions itself)."""
    ops, O, S = _setup()
    from dass_hip.optim import SGD
    from models.deeplab import DeepLab
    from utils.loss import SegmentationLosses

    if engine == "bf16":
        ops.set_compute_dtype(torch.bfloat16)
    else:
        ops.set_f32_mma(engine)
    ncls, n, hw = 19, 2, 65
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=40)
    x, lab = x.cuda(), lab.cuda()
    m1, m2 = O.dropout_masks(n, 3, seed=9)
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    runs = {}
    for name, opt_cls in (("torch", torch.optim.SGD), ("hip", SGD)):
        torch.manual_seed(77)
        pm = DeepLab(backbone="resnet", output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False).cuda()
        opt = opt_cls([{"params": pm.get_1x_lr_params(), "lr": 0.02}, {"params": pm.get_10x_lr_params(), "lr": 0.2}],
                      momentum=0.9, weight_decay=5e-4)
        pm.train()
        losses = []
        for step in range(3):
            opt.zero_grad(set_to_none=True)
            loss = crit(pm(x, dropout_masks=(m1[step].cuda(), m2[step].cuda())), lab)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        # every operand / eval-BN cache of `pm` must describe its CURRENT tensors: a deep copy (new parameter objects, so
        # no cache entry can match) must compute the same logits bit for bit, in train- and in eval-mode BN
        import copy

        fresh = copy.deepcopy(pm)
        with torch.no_grad():
            logits = pm(x, dropout_masks=(m1[0].cuda(), m2[0].cuda())).float().cpu()
            assert torch.equal(logits, fresh(x, dropout_masks=(m1[0].cuda(), m2[0].cuda())).float().cpu()), name
            pm.eval()
            fresh.eval()
            assert torch.equal(pm(x).float().cpu(), fresh(x).float().cpu()), name
        runs[name] = (losses, logits, {k: v.detach().float().cpu().clone() for k, v in pm.named_parameters()})
    lt, lh = runs["torch"][0], runs["hip"][0]
    # large learning rates: a stale step-0 operand in step 1 or 2 moves the loss by O(1)
    tol = 2e-2 if engine == "bf16" else 2e-3
    assert abs(lt[0] - lh[0]) <= 1e-6 * abs(lt[0]) + 1e-6
    # step 2 sees the step-1 update (a stale operand would reproduce the step-1 loss exactly); by step 3 the two
    # optimizers' rounding (fused kernel vs foreach passes) has been amplified by the net, so that bound is loose
    assert abs(lt[1] - lt[0]) > 1e-5 * abs(lt[0]), lt   # (a stale operand repeats the loss EXACTLY; an honest step moved it by 3e-4 here)
    assert abs(lt[1] - lh[1]) <= 0.05 * tol * abs(lt[1]), (lt, lh)
    # step 3: chaotic amplification of the optimizers' last-bit differences reaches percents of the loss on this 65x65 /
    # batch-2 net (seen: 1.8 %); what a stale operand would do -- leave the loss where it was -- is still excluded
    assert abs(lt[2] - lh[2]) <= 0.05 * abs(lt[2]) and abs(lh[2] - lh[1]) > 1e-5 * abs(lh[1]), (lt, lh)
    print(engine, "losses torch", lt, "hip", lh)
    # (the logits of the two trajectories are NOT compared: train-mode BN over 5x5 maps at batch 2 amplifies the optimizers'
    # rounding differences chaotically; the cache-freshness check above is the bit-exact deep-copy comparison)
    # (weights of the two trajectories are not compared either: the update arithmetic itself is checked exactly, on the
    # model's real gradient tensors, by test_hip_sgd_equals_torch_sgd_on_model_gradients below)


def test_bn_eval_cache_sees_running_stat_updates():
    """dass_bn_finalize updates running_mean / running_var through raw pointers: a train-mode forward with NO write to
    the affine parameters in between (frozen affine, BN recalibration) followed by eval() must normalise with the NEW
    running statistics."""
    ops, O, S = _setup()
    import torch.nn as nn

    conv = nn.Conv2d(16, 32, 3, 1, 1, bias=False).cuda()
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
    bn = nn.BatchNorm2d(32).cuda()
    x = (torch.randn(2, 16, 9, 9, generator=torch.Generator().manual_seed(1)) * 3 + 1).cuda()
    ref_bn = nn.BatchNorm2d(32)
    with torch.no_grad():
        bn.eval()
        y0 = ops.conv_bn_act(x, conv, bn, ops.ACT_NONE).float().cpu()       # caches the eval vectors of the initial stats
        for _ in range(3):                                                   # recalibration: train-mode forwards only
            bn.train()
            ops.conv_bn_act(x, conv, bn, ops.ACT_NONE)
            ref_bn.train()
            ref_bn(torch.nn.functional.conv2d(x.cpu(), conv.weight.detach().cpu(), padding=1))
        bn.eval()
        ref_bn.eval()
        y1 = ops.conv_bn_act(x, conv, bn, ops.ACT_NONE).float().cpu()
        want = ref_bn(torch.nn.functional.conv2d(x.cpu(), conv.weight.detach().cpu(), padding=1))
    assert (y1 - y0).abs().max().item() > 1e-2, "running statistics did move"
    assert (y1 - want).abs().max().item() <= 1e-4 * want.abs().max().item()
    # in_scale outside its inference-only path must not be dropped silently
    bn.train()
    with pytest.raises((RuntimeError, AssertionError)):
        ops.conv_bn_act(x, conv, bn, ops.ACT_NONE, in_scale=torch.ones(2, 16, device="cuda"))


def test_get_vote_entropy_for_images_vs_oracle_topk_with_ties():
    """a9 (mc_dropout.py:173-196): loader loop -> T stochastic passes -> vote entropy -> per-image mean over ALL pixels ->
    Python stable sort descending -> first k keys.  Duplicate images in the pool give exactly tied scores: the
    selection must keep the original list order among them, like the reference's sorted(...)."""
    ops, O, S = _setup()
    import constants
    from active_selection.mc_dropout import ActiveSelectionMCDropout

    ncls, hw, T = 19, 65, 6
    from models.deeplab import DeepLab

    om = O.ODeepLab("mobilenet", 16, ncls)
    O.fill_state_dict(om, seed=14)
    pm = DeepLab(backbone="mobilenet", output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False)
    pm.load_state_dict(om.state_dict())
    pm = pm.cuda()
    pm.eval()
    base = [O.synthetic_batch(1, hw, hw, ncls, first_index=800 + i) for i in range(4)]
    order = [0, 1, 0, 2, 1, 3, 0]                                   # images 0 and 1 appear several times: tied scores
    keys = [("img_%03d" % i).encode("ascii") for i in range(len(order))]
    pool = {k: base[j] for k, j in zip(keys, order)}

    class FixedMasks(ActiveSelectionMCDropout):
        """the same dropout masks for every image (what makes duplicates tie), recorded for the oracle"""

        def _votes(self, model, image_batch, steps, masks=None):
            n = image_batch.shape[0]
            m1, m2 = O.dropout_masks(1, steps, seed=33)
            return super()._votes(model, image_batch, steps, masks=(m1.expand(steps, n, 256), m2.expand(steps, n, 256)))

    def factory(images, include_labels, bs=3):
        for i in range(0, len(images), bs):
            chunk = images[i:i + bs]
            yield {"image": torch.cat([pool[k][0] for k in chunk]), "label": torch.cat([pool[k][1] for k in chunk])}

    sel = FixedMasks(ncls, None, hw, 3, loader_factory=factory)
    got = sel.get_vote_entropy_for_images(pm, keys, 4, steps=T)
    assert all(not m.training for m in pm.modules() if isinstance(m, torch.nn.Dropout2d)), "model.eval() on exit"
    # oracle: the same reduction on the CPU model with the same masks
    om.eval()
    m1, m2 = O.dropout_masks(1, T, seed=33)
    scores = []
    for k in keys:
        img, lab = pool[k]
        votes = S.mc_votes(om, img, (m1, m2))
        scores.append(float(S.vote_entropy_maps(votes, lab, ncls)[0].mean()))
    assert scores[0] == scores[2] == scores[6] and scores[1] == scores[4]
    want = S.select_top(scores, keys, 4, reverse=True)
    assert list(got) == list(want), (got, want, scores)
    dev_scores = sel._image_scores(pm, keys, T).cpu().numpy()
    assert np.abs(dev_scores - np.array(scores)).max() <= 1e-3
    assert dev_scores[0] == dev_scores[2] == dev_scores[6], "duplicates must score bit-identically whatever their batch slot"
    # T defaults to constants.MC_STEPS read at call time
    constants.MC_STEPS = 3
    try:
        assert len(sel.get_vote_entropy_for_images(pm, keys[:2], 1)) == 1
    finally:
        constants.MC_STEPS = 20


def test_x3_engine_in_training_matches_classic_engine():
    """DASS_X3=all: forward, input and weight gradients of every dense conv on the pre-split kernels (split rows written by
    the BN passes) against the classic bf16x6 kernels on the same model and batch: same products, same order inside a
    slab -> loss equal to rounding, every gradient within 1e-4 relative (f32 atomics in both weight-gradient kernels)"""
    ops, O, S = _setup()
    from models.deeplab import DeepLab
    from utils.loss import SegmentationLosses

    ncls, n, hw = 19, 2, 65
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=70)
    m1, m2 = O.dropout_masks(n, 1, seed=4)
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    torch.manual_seed(5)
    pm = DeepLab(backbone="resnet", output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False).cuda()
    pm.train()
    pm.freeze_bn()  # running statistics: no batch-statistics amplification in the comparison
    res = {}
    engine = ops.f32_mma()
    ops.set_f32_mma("bf16x6")  # (a property of the six-product kernels: the two engines multiply the same parts in the same order)
    keep = ops.x3_mode()
    try:
        for mode in ("off", "all", "select"):
            ops.set_x3_pipeline(mode)
            pm.zero_grad(set_to_none=True)
            loss = crit(pm(x.cuda(), dropout_masks=(m1[0].cuda(), m2[0].cuda())), lab.cuda())
            loss.backward()
            res[mode] = (loss.item(), {k: p.grad.detach().double().cpu() for k, p in pm.named_parameters()})
    finally:
        ops.set_x3_pipeline(keep)
        ops.set_f32_mma(engine)
    for mode in ("all", "select"):
        assert abs(res["off"][0] - res[mode][0]) <= 1e-6 * abs(res["off"][0])
        worst = max(((res[mode][1][k] - g).norm().item() / max(g.norm().item(), 1e-12), k) for k, g in res["off"][1].items())
        assert worst[0] <= 1e-4, (mode, worst)


def test_hip_sgd_equals_torch_sgd_on_model_gradients():
    """the fused multi-tensor update against torch.optim.SGD on a real model's gradient tensors (arena views for conv
    weights, slices of the BN-sum buffers for gamma / beta, channels_last 1x1 weights): three steps with the SAME gradients
    must move every parameter identically (momentum 0.9, weight decay, two learning-rate groups)"""
    ops, O, S = _setup()
    import copy

    from dass_hip.optim import SGD
    from models.deeplab import DeepLab
    from utils.loss import SegmentationLosses

    ncls, n, hw = 19, 2, 65
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=40)
    torch.manual_seed(3)
    pm = DeepLab(backbone="resnet", output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False).cuda().train()
    SegmentationLosses(cuda=True).build_loss("ce")(pm(x.cuda()), lab.cuda()).backward()
    ref = copy.deepcopy(pm)
    for (k, p), q in zip(pm.named_parameters(), ref.parameters()):
        assert p.grad is not None, k
        q.grad = p.grad.detach().clone()
    init = {k: v.detach().clone() for k, v in pm.named_parameters()}
    groups = lambda m: [{"params": list(m.get_1x_lr_params()), "lr": 0.02}, {"params": list(m.get_10x_lr_params()), "lr": 0.2}]  # noqa: E731
    a, b = SGD(groups(pm), momentum=0.9, weight_decay=5e-4), torch.optim.SGD(groups(ref), momentum=0.9, weight_decay=5e-4)
    for _ in range(3):
        a.step()
        b.step()
    worst = (0.0, None)
    for (k, p), q in zip(pm.named_parameters(), ref.parameters()):
        upd = (q.detach() - init[k]).abs().max().item()
        err = (p.detach() - q.detach()).abs().max().item()
        assert upd > 0, k
        worst = max(worst, (err / upd, k))
    assert worst[0] <= 1e-4, worst  # fma vs separate multiply-add roundings, relative to the size of the 3-step update


def test_deterministic_weight_gradients_are_bit_reproducible():
    """ops.set_deterministic(True): one pixel split per weight-gradient tile -> two backward passes give bit-identical
    gradients for every parameter (the default mode accumulates splits with f32 atomics and may differ in the last bits);
    both modes agree to rounding"""
    ops, O, S = _setup()
    from models.deeplab import DeepLab
    from utils.loss import SegmentationLosses

    ncls, n, hw = 19, 2, 129
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=90)
    torch.manual_seed(9)
    pm = DeepLab(backbone="resnet", output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False).cuda().train()
    pm.freeze_bn()
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    m1, m2 = O.dropout_masks(n, 1, seed=2)

    def grads():
        pm.zero_grad(set_to_none=True)
        crit(pm(x.cuda(), dropout_masks=(m1[0].cuda(), m2[0].cuda())), lab.cuda()).backward()
        return {k: p.grad.detach().clone() for k, p in pm.named_parameters()}

    fast = grads()
    try:
        ops.set_deterministic(True)
        a, b = grads(), grads()
    finally:
        ops.set_deterministic(False)
    assert all(torch.equal(a[k], b[k]) for k in a), [k for k in a if not torch.equal(a[k], b[k])][:5]
    worst = max(((a[k] - fast[k]).norm().item() / max(fast[k].norm().item(), 1e-12), k) for k in a)
    assert worst[0] <= 1e-5, worst


@pytest.mark.parametrize("shape", [(2, 64, 33, 33, 256, 1), (2, 256, 17, 17, 256, 3), (3, 32, 19, 23, 96, 3)])
@pytest.mark.parametrize("residual", [False, True])
def test_bn_sums_path_equals_partial_row_path(shape, residual):
    """train-mode BN through the f64 accumulators (dass_conv2d_igemm_sums / _x3_sums -> dass_bn_apply_train, backward
    dass_bn_bwd_reduce_sums -> dass_bn_bwd_apply_sums) against the partial-row + finalize launches it replaces: same
    f32-inside-a-tile / f64-across arithmetic, so outputs, running statistics and all gradients agree to rounding."""
    from dass_hip import ops

    n, c, h, w, k, ks = shape
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(c, k, ks, padding=ks // 2, bias=False).cuda()
    x0 = torch.randn(n, c, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    res0 = torch.randn(n, k, h, w, device="cuda").contiguous(memory_format=torch.channels_last) if residual else None
    gout = torch.randn(n, k, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    runs = {}
    keep = ops._bn_sum_arena["on"]
    try:
        for on in (False, True):
            ops._bn_sum_arena["on"] = on
            bn = torch.nn.BatchNorm2d(k).cuda()
            torch.manual_seed(11)  # the same affine parameters in both runs
            with torch.no_grad():
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.normal_()
            bn.train()
            conv.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            res = res0.clone().requires_grad_(True) if residual else None
            assert ops.bn_sums_path(bn, k) == on
            out = ops.conv_bn_act(x, conv, bn, ops.ACT_RELU, residual=res)
            out.backward(gout)
            runs[on] = dict(out=out.detach(), rm=bn.running_mean.clone(), rv=bn.running_var.clone(), dx=x.grad, dw=conv.weight.grad,
                            dg=bn.weight.grad, db=bn.bias.grad, dres=res.grad if residual else None)
    finally:
        ops._bn_sum_arena["on"] = keep
    for key, a in runs[False].items():
        if a is None:
            continue
        b = runs[True][key]
        assert (a - b).abs().max().item() <= 2e-6 * max(a.abs().max().item(), 1e-3), key


def test_train_score_train_score_loop_keeps_caches_coherent():
    """the active-learning loop alternates training steps (HIP SGD writes weights and BN running statistics through raw
    pointers) with eval-mode pool scoring (cached split-weight operands, eval-BN vectors, per-tensor split rows, the
    decoder's hoisted first-conv operands).  After every phase change a deep copy of the model -- new parameter objects, no
    cache entry can match -- must produce bit-identical MC-dropout votes and logits."""
    import copy

    ops, O, S = _setup()
    from dass_hip.optim import SGD
    from models.deeplab import DeepLab
    from utils.loss import SegmentationLosses

    ops.set_f32_mma("bf16x6")
    ncls, n, hw, T = 19, 2, 65, 3
    torch.manual_seed(3)
    pm = DeepLab(backbone="resnet", output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False).cuda()
    opt = SGD([{"params": pm.get_1x_lr_params(), "lr": 0.01}, {"params": pm.get_10x_lr_params(), "lr": 0.1}], momentum=0.9, weight_decay=5e-4)
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=77)
    x, lab = x.cuda(), lab.cuda()
    g = torch.Generator().manual_seed(5)
    m1 = (torch.rand(T, n, 256, generator=g) >= 0.5).float() * 2.0
    m2 = (torch.rand(T, n, 256, generator=g) >= 0.1).float() / 0.9
    seen = []
    for phase in range(3):
        pm.train()
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            crit(pm(x), lab).backward()
            opt.step()
        pm.eval()
        fresh = copy.deepcopy(pm)
        with torch.no_grad():
            v = pm.mc_dropout_votes(x, T, masks=(m1, m2))
            assert torch.equal(v, fresh.mc_dropout_votes(x, T, masks=(m1, m2))), phase
            lg = pm(x).float()
            assert torch.equal(lg, fresh(x).float()), phase
        seen.append(lg.cpu())
    assert not torch.equal(seen[0], seen[2]), "the training steps must move the model for this test to mean anything"
