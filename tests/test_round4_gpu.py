"""Round 4: all T stochastic passes of a scoring batch as one launch per conv (decoder.head_mc_all), against the per-pass tail and
against T full forwards with the same masks."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(backbone, seed):
    from dass_hip import ops
    from models.deeplab import DeepLab
    from oracle import deeplab_cpu as O

    ops.set_compute_dtype(torch.float32)
    om = O.ODeepLab(backbone, 16, 19)
    O.fill_state_dict(om, seed=seed)
    pm = DeepLab(backbone=backbone, output_stride=16, num_classes=19, sync_bn=False, pretrained=False)
    pm.load_state_dict(om.state_dict())
    return pm.cuda().eval(), O


@pytest.mark.parametrize("backbone,n,hw,T", [("resnet", 3, 129, 4), ("mobilenet", 2, 97, 5), ("resnet", 8, 257, 10)])
def test_batched_mc_tail_equals_per_pass_tail(backbone, n, hw, T):
    """DASS_MC_BATCHED (default): N x T (image, pass) pairs ride as the "images" of ONE per-image conv launch, sharing the batch's
    deterministic residual (dass_conv2d_x3_per_image_rep, dass_split3_rows_packed_rep).  Same products per output as the per-pass
    launches, summed in a different order where the schedules differ (whole tiles vs stream-K): votes may move only at exact
    near-ties; against T full forwards with the same masks the argmax agrees wherever the top-2 margin exceeds 1e-3."""
    from dass_hip import ops

    keep_mma, keep_env = ops.f32_mma(), os.environ.get("DASS_MC_BATCHED")
    try:
        ops.set_f32_mma("f16x3")
        pm, O = _model(backbone, seed=11)
        x, _ = O.synthetic_batch(n, hw, hw, 19, first_index=300)
        m1, m2 = O.dropout_masks(n, T, seed=12)
        xd = x.cuda()
        state = pm.mc_prefix(xd)
        os.environ["DASS_MC_BATCHED"] = "1"
        vb = pm.mc_tail(state, T, masks=(m1, m2))
        assert torch.equal(vb, pm.mc_tail(state, T, masks=(m1, m2)))   # deterministic
        os.environ["DASS_MC_BATCHED"] = "0"
        vs = pm.mc_tail(state, T, masks=(m1, m2))
        diff = int((vb != vs).sum())
        print("%s %dx%d T=%d: batched vs per-pass vote differences %d of %d" % (backbone, n, hw, T, diff, vb.numel()))
        assert vb.shape == (n, T, hw, hw) and diff <= 1e-5 * vb.numel() + 2
        with torch.no_grad():
            for t in (0, T - 1):
                full = pm(xd, dropout_masks=(m1[t].cuda(), m2[t].cuda()))
                top = full.topk(2, dim=1)[0]
                safe = (top[:, 0] - top[:, 1]) > 1e-3
                assert torch.equal(full.argmax(1)[safe].to(torch.uint8), vb[:, t][safe])
    finally:
        ops.set_f32_mma(keep_mma)
        if keep_env is None:
            os.environ.pop("DASS_MC_BATCHED", None)
        else:
            os.environ["DASS_MC_BATCHED"] = keep_env


@pytest.mark.parametrize("gates", [False, True])
@pytest.mark.parametrize("with_res", [False, True])
@pytest.mark.parametrize("want_dx32", [False, True])
def test_lean_bn_backward_kernel_equals_general_kernel(gates, with_res, want_dx32):
    """bn_bwd_fast_kernel (the train step's form of dass_bn_bwd_apply_sums) against the general kernel (DASS_BN_BWD_FAST=0, read
    per call) on the same arguments: identical f32 rows, split rows, residual gradient and parameter gradients, bit for bit; and
    against the formula in f64."""
    from dass_hip import ops
    from dass_hip._lib import check, lib

    keep, keep_env = ops.f32_mma(), os.environ.get("DASS_BN_BWD_FAST")
    try:
        ops.set_f32_mma("f16x3")
        torch.manual_seed(7)
        m, k = 8 * 33 * 33, 256
        dev = "cuda"
        x = torch.randn((m, k), device=dev)                      # conv output (pre-BN)
        dout = torch.randn((m, k), device=dev) * 1e-3
        gamma = torch.rand((k,), device=dev) + 0.5
        mean, var = x.mean(0), x.var(0, unbiased=False)
        invstd = (var + 1e-5).rsqrt()
        scale, shift = (gamma * invstd).contiguous(), (-mean * gamma * invstd).contiguous()
        res = torch.randn((m, k), device=dev) if with_res else None
        pre = torch.addcmul(shift.expand(m, k), x, scale.expand(m, k))
        gate = (pre + res > 0) if with_res else None             # residual layers: the gate exists only as stored bits
        use_bits = gates or with_res
        gbits = None
        if use_bits:
            gt = gate if gate is not None else pre > 0
            g4 = gt.view(m, k // 4, 4).to(torch.uint8)
            gbits = (g4[..., 0] | (g4[..., 1] << 1) | (g4[..., 2] << 2) | (g4[..., 3] << 3)).contiguous()
        sums = torch.zeros((3 * k,), dtype=torch.float64, device=dev)

        def run(fast):
            os.environ["DASS_BN_BWD_FAST"] = "1" if fast else "0"
            dx = torch.empty((m, k), device=dev) if want_dx32 else None
            dres = torch.empty((m, k), device=dev) if with_res else None
            dx3 = ops.x3_alloc_for(m, k, dev)
            pg = torch.empty((2, k), device=dev)
            check(lib.dass_bn_bwd_apply_sums(ops._p(dout), k, None, k, ops._p(x), k, ops._p(mean), ops._p(invstd), ops._p(gamma), ops._p(sums),
                                             ops._p(pg[0]), ops._p(pg[1]), ops._p(None if use_bits else scale), ops._p(None if use_bits else shift), None,
                                             ops._p(dx), k, ops._p(dres), k, m, k, 33 * 33, float(m), ops.ACT_RELU, ops._p(gbits),
                                             gbits.numel() if use_bits else 0, ops.F32, ops._p(dx3), ops._stream()), "bn_bwd_apply_sums")
            return dx, dres, dx3, pg

        # the sums the reduce pass would have left (sum dz, sum dz xhat, max |dz| per channel), with the gate the kernels will use
        gt = (gate if gate is not None else pre > 0)
        dz = dout * gt
        xh = ((x - mean) * invstd)
        sums[:k] = dz.double().sum(0)
        sums[k:2 * k] = (dz.double() * xh.double()).sum(0)
        sums.view(torch.float32)[4 * k:5 * k] = dz.abs().amax(0)
        a, b = run(True), run(False)
        assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
        if want_dx32:
            assert torch.equal(a[0], b[0])
            ref = (dz.double() - (sums[:k] + xh.double() * sums[k:2 * k]) / m) * (gamma * invstd).double()
            near = (pre.abs() < 1e-6) if not use_bits else torch.zeros_like(pre, dtype=torch.bool)   # (re-derived gates: fma vs two roundings)
            err = ((a[0].double() - ref).abs() * (~near)).max().item()
            assert err <= 2e-6 * ref.abs().max().item() + 1e-12, err
        if with_res:
            assert torch.equal(a[1], b[1])
    finally:
        ops.set_f32_mma(keep)
        if keep_env is None:
            os.environ.pop("DASS_BN_BWD_FAST", None)
        else:
            os.environ["DASS_BN_BWD_FAST"] = keep_env


def test_merged_loader_batches_give_the_same_features_and_scores():
    """DASS_SCORE_MERGE: two loader batches per scoring forward.  Core-set features (no randomness) of a 10-image pool with merge 1
    and merge 2 agree to f32 rounding of a different tile schedule (<= 2e-5 of the feature scale); the MC-dropout scores of the merged
    run stay deterministic under a fixed seed and rank the pool like the unmerged run's draws would only by chance -- so here
    only shape, finiteness and determinism are asserted for them."""
    from active_selection.core_set import ActiveSelectionCoreSet
    from active_selection.mc_dropout import ActiveSelectionMCDropout, _turn_on_dropout
    from dass_hip.dist import ModuleWrapper

    pm, O = _model("resnet", seed=15)
    hw = 513   # (the 2736-wide core-set feature = 304 channels x 3 x 3 pooled cells of the 129 x 129 decoder map)
    pool = {("img_%03d" % i).encode("ascii"): O.synthetic_batch(1, hw, hw, 19, first_index=500 + i) for i in range(10)}
    keys = list(pool)

    def fimg(images, include_labels, bs=3):
        for i in range(0, len(images), bs):
            yield torch.cat([pool[k][0] for k in images[i:i + bs]])

    def fdict(images, include_labels, bs=3):
        for i in range(0, len(images), bs):
            chunk = images[i:i + bs]
            yield {"image": torch.cat([pool[k][0] for k in chunk]), "label": torch.cat([pool[k][1] for k in chunk])}

    keep = os.environ.get("DASS_SCORE_MERGE")
    try:
        feats = {}
        for merge in ("1", "2"):
            os.environ["DASS_SCORE_MERGE"] = merge
            feats[merge] = ActiveSelectionCoreSet(None, hw, 3, loader_factory=fimg)._features(ModuleWrapper(pm), keys).cpu()
        assert feats["1"].shape == feats["2"].shape == (10, 2736)
        assert (feats["1"] - feats["2"]).abs().max().item() <= 2e-5 * feats["1"].abs().max().item()
        os.environ["DASS_SCORE_MERGE"] = "2"
        sel = ActiveSelectionMCDropout(19, None, hw, 3, loader_factory=fdict)
        pm.apply(_turn_on_dropout)
        runs = []
        for _ in range(2):
            torch.manual_seed(5)
            torch.cuda.manual_seed(5)
            runs.append(sel._image_scores(pm, keys, 4).cpu())
        pm.eval()
        assert runs[0].shape == (10,) and torch.isfinite(runs[0]).all() and torch.equal(runs[0], runs[1])
    finally:
        if keep is None:
            os.environ.pop("DASS_SCORE_MERGE", None)
        else:
            os.environ["DASS_SCORE_MERGE"] = keep


def test_rows_only_outputs_change_nothing_but_the_bytes_written():
    """conv_bn_act(sole_consumer=True): conv1 / conv2 of every bottleneck write ONLY the split rows their one reader takes.  One
    train step of DeepLab-ResNet50 with the switch on and off (deterministic weight gradients): identical loss, identical gradients
    of all parameters, identical running statistics; eval logits identical; and a rows-only activation handed to a consumer that
    does not take split rows fails loudly instead of reading memory nobody wrote."""
    from dass_hip import ops
    from models.deeplab import DeepLab
    from utils.loss import SegmentationLosses

    keep = ops.f32_mma()
    try:
        ops.set_f32_mma("f16x3")
        ops.set_deterministic(True)
        _, O = _model("resnet", seed=3)
        om = O.ODeepLab("resnet", 16, 19)
        O.fill_state_dict(om, seed=3, randomize_bn_stats=False)
        x, lab = O.synthetic_batch(2, 129, 129, 19, first_index=700)
        m1, m2 = O.dropout_masks(2, 1, seed=4)
        res = {}
        for on in (True, False):
            ops.set_rows_only(on)
            pm = DeepLab(backbone="resnet", output_stride=16, num_classes=19, sync_bn=False, pretrained=False)
            pm.load_state_dict(om.state_dict())
            pm = pm.cuda().train()
            loss = SegmentationLosses(cuda=True).build_loss("ce")(pm(x.cuda(), dropout_masks=(m1[0].cuda(), m2[0].cuda())), lab.cuda())
            loss.backward()
            grads = {k: p.grad.clone() for k, p in pm.named_parameters()}
            stats = {k: v.clone() for k, v in pm.state_dict().items() if "running" in k}
            pm.eval()
            with torch.no_grad():
                logits = pm(x.cuda()).clone()
            res[on] = (loss.item(), grads, stats, logits)
        assert res[True][0] == res[False][0]
        for k in res[True][1]:
            assert torch.equal(res[True][1][k], res[False][1][k]), k
        for k in res[True][2]:
            assert torch.equal(res[True][2][k], res[False][2][k]), k
        assert torch.equal(res[True][3], res[False][3])
        # misuse: the producer was told its only reader is `c2`, but the tensor goes to a depthwise conv
        ops.set_rows_only(True)
        c1 = torch.nn.Conv2d(64, 64, 1, bias=False).cuda()
        c2 = torch.nn.Conv2d(64, 64, 3, padding=1, bias=False).cuda()
        dw = torch.nn.Conv2d(64, 64, 3, padding=1, groups=64, bias=False).cuda()
        xin = torch.randn(2, 64, 33, 33, device="cuda").contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            o = ops.conv_bn_act(xin, c1, None, ops.ACT_RELU, consumer=c2, sole_consumer=True)
            assert o.__dict__.get("_dass_rows_only")
            ops.conv_bn_act(o, c2, None, ops.ACT_RELU)           # the declared reader: fine
            with pytest.raises(RuntimeError, match="sole_consumer"):
                ops.conv_bn_act(o, dw, None, ops.ACT_NONE)
    finally:
        ops.set_rows_only(True)
        ops.set_deterministic(False)
        ops.set_f32_mma(keep)


@pytest.mark.parametrize("engine", ["f16x3", "bf16x1"])
def test_graphed_train_step_follows_the_eager_trajectory(engine):
    """dass_hip.graph.GraphedStep: zero_grad + forward + CE + backward + SGD captured into one hipGraph.  From the same initial state,
    2 eager warm-up steps + 4 replays give the losses of 6 eager steps (f32 atomics reorder the weight-gradient sums: 1e-4), the
    running statistics and weights after the run agree, and an eval forward after the replays sees the UPDATED weights (cached
    weight operands are invalidated by the replay)."""
    from dass_hip import ops
    from dass_hip.graph import GraphedStep
    from dass_hip.optim import SGD
    from models.deeplab import DeepLab
    from utils.loss import SegmentationLosses

    keep = ops.f32_mma()
    try:
        ops.set_f32_mma(engine)
        _, O = _model("resnet", seed=3)
        om = O.ODeepLab("resnet", 16, 19)
        O.fill_state_dict(om, seed=8, randomize_bn_stats=False)
        x, lab = O.synthetic_batch(4, 129, 129, 19, first_index=800)
        xd, ld = x.cuda(), lab.cuda()
        crit = SegmentationLosses(cuda=True).build_loss("ce")
        runs = {}
        for mode in ("eager", "eager2", "graph"):
            pm = DeepLab(backbone="resnet", output_stride=16, num_classes=19, sync_bn=False, pretrained=False)
            pm.load_state_dict(om.state_dict())
            pm = pm.cuda().train()
            opt = SGD([{"params": pm.get_1x_lr_params(), "lr": 0.01}, {"params": pm.get_10x_lr_params(), "lr": 0.1}], momentum=0.9, weight_decay=5e-4)
            torch.manual_seed(0)
            torch.cuda.manual_seed(0)
            masks = O.dropout_masks(4, 1, seed=9)
            dm = (masks[0][0].cuda(), masks[1][0].cuda())    # fixed Dropout2d masks: the two runs must draw nothing

            def step():
                opt.zero_grad(set_to_none=True)
                loss = crit(pm(xd, dropout_masks=dm), ld)
                loss.backward()
                opt.step()
                return loss

            losses = []
            if mode != "graph":
                for _ in range(6):
                    losses.append(float(step().detach()))
            else:
                gs = GraphedStep(step, warmup=2)
                for _ in range(4):
                    losses.append(float(gs().detach()))
            pm.eval()
            with torch.no_grad():
                logits = pm(xd).float().cpu()
                # the sharp check for stale operand caches after replays (which step the optimizer behind autograd's back): a deep copy has
                # new parameter objects, no cache entry can match them, and it must compute the same logits bit for bit
                import copy

                fresh = copy.deepcopy(pm).eval()
                assert torch.equal(logits, fresh(xd).float().cpu()), mode
            runs[mode] = (losses, {k: v.detach().float().cpu().clone() for k, v in pm.state_dict().items()}, logits)
        le, lg = runs["eager"][0], runs["graph"][0]
        print(engine, "eager", ["%.5f" % v for v in le], "graph", ["%.5f" % v for v in lg])
        assert all(abs(a - b) <= 2e-3 * abs(a) for a, b in zip(le[2:], lg)), (le, lg)
        assert lg[-1] < lg[0]
        sa, sb = runs["eager"][1], runs["graph"][1]
        for k in sa:
            if "num_batches_tracked" in k:
                assert int(sb[k]) == int(sa[k]) == 6, (k, sa[k], sb[k])
            elif "running" in k and "bn_global_average_pool" not in k:
                # (the ASPP image-pool BN -- 4 samples per channel -- amplifies the reordered f32 atomics of the weight gradients ~1e3 and is
                #  left out; bf16 products amplify them everywhere else too)
                # (129^2 crops: the 9x9 ASPP maps hold 324 samples per channel, and a variance over so few moves by a few 1e-2 between two
                #  runs of the SAME eager step already -- the weight gradients' f32 atomics land in a different order each run)
                tol = 5e-2
                assert (sa[k] - sb[k]).abs().max().item() <= tol * max(1.0, sa[k].abs().max().item()), k
        # eval after the replays runs on the updated weights (stale operand caches would be off by a step).  The yardstick is what two
        # runs of the SAME eager loop differ by after six steps (f32 atomics of the weight gradients in a different order each run)
        # (RMS over all logits: the maximum is set by a handful of near-tie pixels of the 129^2 crops and moves by 2-4x between runs)
        ref = runs["eager"][2]
        rms = lambda t: float(t.double().pow(2).mean().sqrt())
        dl = rms(ref - runs["graph"][2]) / rms(ref)
        noise = rms(ref - runs["eager2"][2]) / rms(ref)
        dmax = (ref - runs["graph"][2]).abs().max().item() / ref.abs().max().item()
        print(engine, "eval logits: graph vs eager rms %.3e (max %.3e), eager vs eager rms %.3e" % (dl, dmax, noise))
        # (trajectory-level bound only: two eager runs of this 129^2 / batch-4 step differ by 1e-3 ... 1e-2 (f16x3) and 3e-2 ... 1.5e-1 (bf16x1) in RMS)
        assert dl <= max(3e-2 if engine == "f16x3" else 1e-1, 4 * noise), (dl, noise)
    finally:
        ops.set_f32_mma(keep)
