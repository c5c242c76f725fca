"""End-to-end parity of the HIP product path (through the reference's surface: models.deeplab.DeepLab,
utils.loss.SegmentationLosses, active_selection.*) against
  (a) the golden fixtures produced by the REFERENCE's own modules (tests/golden, oracle/make_goldens.py),
  (b) the CPU oracle run on the same seeded inputs.
Tolerances follow BASELINE.json's north_star: logits and entropy scores within 1e-3 (f32), argmax label
maps bit-exact wherever the reference's own top-2 margin exceeds 1e-3 (nearer ties are reported).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _setup():
    from dass_hip import ops

    ops.set_compute_dtype(torch.float32)
    from oracle import deeplab_cpu as O
    from oracle import selection_cpu as S

    return ops, O, S


def _product(backbone, ncls, O, seed, randomize_bn_stats=True):
    from models.deeplab import DeepLab

    om = O.ODeepLab(backbone, 16, ncls)
    O.fill_state_dict(om, seed=seed, randomize_bn_stats=randomize_bn_stats)
    pm = DeepLab(backbone=backbone, output_stride=16, num_classes=ncls, sync_bn=False, freeze_bn=False, pretrained=False)
    assert list(pm.state_dict().keys()) == list(om.state_dict().keys()), "state_dict keys must match the reference's"
    pm.load_state_dict(om.state_dict())
    return pm.cuda(), om


@pytest.fixture(autouse=True)
def _restore_mma_mode():
    from dass_hip import ops

    mode = ops.f32_mma()
    yield
    ops.set_f32_mma(mode)


@pytest.mark.parametrize("engine", ["f16x3", "bf16x6", "f32"])
@pytest.mark.parametrize("tag,backbone", [("mobilenet", "mobilenet"), ("resnet50", "resnet"), ("resnet101", "resnet101"),
                                          ("mobilenet_voc", "mobilenet")])
def test_e2e_logits_vs_reference_golden(tag, backbone, engine):
    """the three parity-grade conv engines (two scaled f16 parts / three products -- the default; exact three-way bf16 split / six
    products; plain f32 MFMA) against the reference's own logits"""
    ops, O, S = _setup()
    ops.set_f32_mma(engine)
    g = np.load(os.path.join(GOLD, "e2e_%s.npz" % tag))
    n, hw, ncls = [int(v) for v in g["meta"]]
    pm, _ = _product(backbone, ncls, O, seed=1)
    pm.eval()
    x, _ = O.synthetic_batch(n, hw, hw, ncls)
    pm.set_return_features(True)
    with torch.no_grad():
        out, feats = pm(x.cuda())
    assert out.shape == (n, ncls, hw, hw) and out.dtype == torch.float32 and out.is_contiguous()
    ref = torch.from_numpy(g["logits"])
    err = (out.cpu() - ref).abs().max().item()
    assert err <= 1e-3, "logits max abs err %.3e" % err
    margin = torch.from_numpy(g["margin"].astype(np.float32))
    safe = margin > 1e-3
    am = out.argmax(1).cpu()
    ref_am = torch.from_numpy(g["argmax"]).long()
    assert torch.equal(am[safe], ref_am[safe]), "argmax differs outside near-ties"
    flips = int((am != ref_am).sum())
    print("%s: logits err %.2e, near-tie pixels %d, argmax flips %d" % (tag, err, int((~safe).sum()), flips))
    assert flips <= int((~safe).sum())
    pooled = torch.nn.functional.avg_pool2d(feats.float().cpu(), 8, 4)
    assert (pooled - torch.from_numpy(g["feat_pooled"])).abs().max().item() <= 1e-3


@pytest.mark.parametrize("engine", ["f16x3", "bf16x6", "f32"])
def test_mc_dropout_votes_and_entropy_vs_reference_golden(engine):
    ops, O, S = _setup()
    ops.set_f32_mma(engine)
    g = np.load(os.path.join(GOLD, "mc_dropout_mobilenet.npz"))
    n, hw, ncls, T = [int(v) for v in g["meta"]]
    pm, om = _product("mobilenet", ncls, O, seed=2)
    pm.eval()
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=50)
    m1, m2 = O.dropout_masks(n, T, seed=3)
    ref_votes = torch.from_numpy(g["votes"]).long()
    # margins of the reference passes decide where exactness is demanded
    om.eval()
    with torch.no_grad():
        margins = []
        for t in range(T):
            top = om(x, (m1[t], m2[t])).topk(2, dim=1)[0]
            margins.append(top[:, 0] - top[:, 1])
    safe = torch.stack(margins, 1) > 1e-3
    # (1) hoisted fast path
    votes = pm.mc_dropout_votes(x.cuda(), T, masks=(m1, m2)).cpu().long()
    assert torch.equal(votes[safe], ref_votes[safe])
    assert int((votes != ref_votes).sum()) <= int((~safe).sum())
    # (2) the reference way: T full forwards with the same masks
    with torch.no_grad():
        full = torch.stack([pm(x.cuda(), dropout_masks=(m1[t].cuda(), m2[t].cuda())).argmax(1) for t in range(T)], 1).cpu()
    assert torch.equal(full[safe], ref_votes[safe])
    # (3) reduction kernel on the reference's own votes -> the reference's entropy maps / means
    emap, means = ops.vote_entropy(torch.from_numpy(g["votes"]).cuda(), lab.cuda(), ncls)
    assert (emap.cpu() - torch.from_numpy(g["entropy"])).abs().max().item() <= 1e-5
    assert (means.cpu() - torch.from_numpy(g["mean"])).abs().max().item() <= 1e-5
    # (4) end to end through the selector surface, scores within 1e-3 of the reference's
    from active_selection.mc_dropout import ActiveSelectionMCDropout

    sel = ActiveSelectionMCDropout(ncls, None, hw, n)
    maps, mean2 = sel._get_vote_entropy_for_batch(pm, x.cuda(), lab.cuda(), steps=T, masks=(m1, m2), with_means=True)
    assert len(maps) == n and maps[0].shape == (hw, hw)
    if int((votes != ref_votes).sum()) == 0:
        assert (mean2.cpu() - torch.from_numpy(g["mean"])).abs().max().item() <= 1e-3


@pytest.mark.parametrize("T,C", [(10, 19), (20, 21)])
def test_vote_entropy_scripted_golden(T, C):
    ops, O, S = _setup()
    g = np.load(os.path.join(GOLD, "vote_entropy_T%d_C%d.npz" % (T, C)))
    emap, means = ops.vote_entropy(torch.from_numpy(g["votes"]).cuda(), torch.from_numpy(g["label"]).cuda(), C)
    ref = torch.from_numpy(g["entropy"])
    assert (emap.cpu() - ref).abs().max().item() <= 1e-5
    assert (means.cpu() - ref.mean(dim=(1, 2))).abs().max().item() <= 1e-5


def test_losses_vs_reference_golden():
    ops, O, S = _setup()
    from utils.loss import SegmentationLosses

    g = np.load(os.path.join(GOLD, "loss.npz"))
    gen = torch.Generator().manual_seed(7)
    logit = torch.randn(2, 19, 21, 23, generator=gen) * 3
    target = torch.randint(0, 19, (2, 21, 23), generator=gen).float()
    target[:, :3] = 255
    wt = torch.rand(19, generator=gen) + 0.5
    for wname, w in (("plain", None), ("weighted", wt)):
        crit = SegmentationLosses(weight=w, cuda=True)
        for mode in ("ce", "focal"):
            lg = logit.cuda().requires_grad_(True)
            loss = crit.build_loss(mode)(lg, target.cuda())
            loss.backward()
            ref = float(g["%s_%s_loss" % (mode, wname)])
            assert abs(loss.item() - ref) <= 1e-5 * max(1.0, abs(ref)), (mode, wname, loss.item(), ref)
            gref = torch.from_numpy(g["%s_%s_grad" % (mode, wname)])
            assert (lg.grad.cpu() - gref).abs().max().item() <= 1e-4 * max(gref.abs().max().item(), 1e-6)
    with pytest.raises(NotImplementedError):
        SegmentationLosses().build_loss("dice")
    sw = torch.tensor([1.0, 0.25])
    val = SegmentationLosses(cuda=True).SampleWeightedCrossEntropyLoss(logit.cuda(), target.cuda(), sw)
    assert abs(val.item() - float(g["sample_weighted_loss"])) <= 1e-5


def test_ceal_scores_vs_golden_and_selector():
    ops, O, S = _setup()
    g = np.load(os.path.join(GOLD, "softmax_scores.npz"))
    gen = torch.Generator().manual_seed(8)
    logits = torch.randn(2, 19, 17, 19, generator=gen) * 2
    lab = torch.randint(0, 19, (2, 17, 19), generator=gen).float()
    lab[:, :2] = 255
    for mode, key in ((0, "conf"), (1, "margin"), (2, "entropy")):
        smap, mean = ops.softmax_scores(logits.cuda(), lab.cuda(), 19, mode, want_map=True)
        ref = torch.from_numpy(g[key])
        assert (smap.cpu() - ref).abs().max().item() <= 1e-5, key
        assert (mean.cpu() - ref.mean(dim=(1, 2))).abs().max().item() <= 1e-5
    assert np.array_equal(ops.weak_labels(logits.cuda(), lab.cuda(), 19).cpu().numpy(), g["weak"])

    # selector surface on a synthetic pool vs the oracle computing the same scores on the CPU
    from active_selection.ceal import ActiveSelectionCEAL

    pm, om = _product("mobilenet", 19, O, seed=6)
    pm.eval()
    om.eval()
    keys = [("img_%03d" % i).encode("ascii") for i in range(5)]
    pool = {k: O.synthetic_batch(1, 65, 65, 19, first_index=300 + i) for i, k in enumerate(keys)}

    def factory(images, include_labels, bs=2):
        for i in range(0, len(images), bs):
            chunk = images[i:i + bs]
            yield {"image": torch.cat([pool[k][0] for k in chunk]), "label": torch.cat([pool[k][1] for k in chunk])}

    sel = ActiveSelectionCEAL(19, None, 65, 2, loader_factory=factory)
    with torch.no_grad():
        lo = torch.cat([om(pool[k][0]) for k in keys])
    labs = torch.cat([pool[k][1] for k in keys])
    conf, margin, ent = S.softmax_score_maps(lo, labs, 19)
    ent_scores = [float(np.mean(e.numpy())) for e in ent]
    got_sel, got_ent = sel.get_maximum_entropy_samples(pm, keys, 2)
    assert np.abs(np.array(got_ent) - np.array(ent_scores)).max() <= 1e-3
    assert list(got_sel) == S.select_top(ent_scores, keys, 2, reverse=True)
    conf_scores = [float(torch.mean(c)) for c in conf]
    assert list(sel.get_least_confident_samples(pm, keys, 2)) == S.select_top(conf_scores, keys, 2, reverse=False)
    weak = sel.get_weakly_labeled_data(pm, keys, threshold=max(ent_scores) + 1, entropies=None)
    assert set(weak.keys()) == set(keys) and weak[keys[0]].dtype == np.uint8 and weak[keys[0]].shape == (65, 65)


def test_coreset_selector_vs_oracle():
    ops, O, S = _setup()
    from active_selection.core_set import ActiveSelectionCoreSet

    g = np.load(os.path.join(GOLD, "kcenter.npz"))
    cs = ActiveSelectionCoreSet(None, None, None)
    assert cs._select_batch(g["small_feats"].astype(np.float32), [6], 5) == g["small_picks"].tolist() == [0, 2, 8, 4, 7]
    big = np.asarray(O._hash_uniform(300 * 2736, 99), dtype=np.float32).reshape(300, 2736)
    assert cs._select_batch(big, list(range(10)), 25) == g["big_picks"].tolist()

    pm, om = _product("mobilenet", 19, O, seed=9)
    pm.eval()
    om.eval()
    keys = [("img_%03d" % i).encode("ascii") for i in range(5)]
    pool = {k: O.synthetic_batch(1, 513, 513, 19, first_index=400 + i)[0] for i, k in enumerate(keys)}

    def factory(images, include_labels, bs=2):
        for i in range(0, len(images), bs):
            yield torch.cat([pool[k] for k in images[i:i + bs]])

    class Wrapper(torch.nn.Module):  # selectors receive a DataParallel-style wrapper with .module
        def __init__(self, m):
            super().__init__()
            self.module = m

        def forward(self, x):
            return self.module(x)

    sel = ActiveSelectionCoreSet(None, 513, 2, loader_factory=factory)
    got = sel.get_k_center_greedy_selections(2, Wrapper(pm), keys[2:], keys[:2])
    om.return_features = True
    with torch.no_grad():
        feats = torch.cat([om(pool[k])[1] for k in keys])
    f64 = S.coreset_features(feats, 64)
    assert f64.shape == (5, 2736)
    got_feats = sel._features(Wrapper(pm), keys).cpu().numpy()
    assert np.abs(got_feats - f64).max() <= 1e-3
    picks, _ = S.kcenter_greedy(f64, [0, 1], 2)
    assert got == [keys[i] for i in picks]
    assert pm.return_features is False


def test_region_selection_vs_reference_golden():
    ops, O, S = _setup()
    from active_selection.mc_dropout import ActiveSelectionMCDropout

    g = np.load(os.path.join(GOLD, "nms_png.npz"))
    imgs = torch.stack([torch.from_numpy(g["img0"].astype(np.float32) / 256), torch.from_numpy(g["img1"].astype(np.float32) / 256)])
    maps = ops.box_sum(imgs.cuda(), 127)
    ops.minmax_normalize_(maps)
    regions, count = ActiveSelectionMCDropout.square_nms(maps, 127, (512 * 512) // (127 * 127))
    assert count == int(g["count"])
    assert regions[0] == [tuple(r) for r in g["regions0"].tolist()]
    assert regions[1] == [tuple(r) for r in g["regions1"].tolist()]
    em = torch.rand(40, 40)
    ref = S.suppress_labeled(em.clone(), [(3, 4, 10, 12), (30, 30, 10, 10)])
    emd = em.clone().cuda()
    ActiveSelectionMCDropout.suppress_labeled_entropy(emd, [(3, 4, 10, 12), (30, 30, 10, 10)])
    assert torch.equal(emd.cpu(), ref)


def test_train_two_steps_vs_reference_golden():
    """forward (train-mode BN) + CE + backward + SGD, twice, against the reference's recorded losses and
    weights (G12) and against the oracle's first-step gradients."""
    ops, O, S = _setup()
    from utils.loss import SegmentationLosses

    g = np.load(os.path.join(GOLD, "train2_mobilenet.npz"))
    ncls, n, hw = 19, 2, 97
    pm, om = _product("mobilenet", ncls, O, seed=4, randomize_bn_stats=False)
    pm.train()
    om.train()
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=200)
    m1, m2 = O.dropout_masks(n, 2, seed=5)
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    lr = 0.01
    opt = torch.optim.SGD([{"params": pm.get_1x_lr_params(), "lr": lr}, {"params": pm.get_10x_lr_params(), "lr": lr * 10}],
                          momentum=0.9, weight_decay=5e-4, nesterov=False)
    # oracle first-step gradients
    lo = S.ce_loss(om(x, (m1[0], m2[0])), lab)
    lo.backward()
    ograd = {k: p.grad.clone() for k, p in om.named_parameters()}
    losses = []
    for step in range(2):
        opt.zero_grad()
        loss = crit(pm(x.cuda(), dropout_masks=(m1[step].cuda(), m2[step].cuda())), lab.cuda())
        loss.backward()
        if step == 0:
            # train-mode BN over the deepest 7x7 maps (mostly fixed_padding zeros) has near-zero-variance
            # channels whose invstd ~ 1/sqrt(eps) amplifies f32 summation-order noise: the reference and
            # the oracle (both stock PyTorch) already differ by ~1e-2 of the update there
            # (oracle/make_goldens.py), so the bound is on the relative L2 error per tensor.
            rels = []
            # parameters whose true gradient is exactly zero (a per-channel shift in front of a train-mode BN,
            # e.g. aspp.bn_global_average_pool.bias) hold pure rounding noise: floor the denominator
            floor = 1e-3 * float(np.median([v.norm().item() for v in ograd.values()]))
            for k, p in pm.named_parameters():
                ref = ograd[k]
                rels.append(((p.grad.cpu() - ref).norm().item() / max(ref.norm().item(), floor), k))
            rels.sort(reverse=True)
            print("first-step gradient rel-L2 err vs oracle, worst 3:", rels[:3])
            assert rels[0][0] <= 5e-2, rels[:3]
            # measured noise floor of stock PyTorch itself on this net (f32 vs f64 oracle): worst 3e-2, median 5e-3
            assert np.median([r for r, _ in rels]) <= 2e-2
        opt.step()
        losses.append(loss.item())
    ref_losses = g["losses"]
    assert abs(losses[0] - ref_losses[0]) <= 1e-4 * abs(ref_losses[0]), (losses, ref_losses)
    assert abs(losses[1] - ref_losses[1]) <= 5e-3 * abs(ref_losses[1]), (losses, ref_losses)
    sd = pm.state_dict()
    for key in g.files:
        if key == "losses" or key.startswith("init__"):
            continue
        name = key.replace("__", ".")
        got = sd[name].detach().float().cpu().reshape(-1)[:4096]
        ref = torch.from_numpy(g[key])
        init = torch.from_numpy(g["init__" + key])
        upd = max((ref - init).abs().max().item(), 1e-3)
        rel = (got - ref).abs().max().item() / upd
        # The 2-step trajectory is chaotic in f32: stock PyTorch itself moves these slices by 0.30 of the update
        # between 1 and 8 threads and by 0.10-0.23 between f32 and f64 (measured with oracle/),
        # and its step-2 loss by 1.4e-3.  The tight checks are the step-1 loss and the step-1 gradients above.
        assert rel <= 0.35, "%s differs from the reference after 2 SGD steps: %.3e of the update" % (name, rel)
    assert int(sd["decoder.bn1.num_batches_tracked"]) == 2


@pytest.mark.parametrize("backbone,hw", [("mobilenet", 65), ("resnet", 65)])
def test_backward_frozen_bn_vs_oracle(backbone, hw):
    """freeze_bn() training (BN on running stats, gamma/beta still trained -- deeplab.py:64-69): without the
    batch-statistics amplification every gradient must match the CPU oracle tightly."""
    ops, O, S = _setup()
    from utils.loss import SegmentationLosses

    ncls, n = 19, 2
    pm, om = _product(backbone, ncls, O, seed=21)
    pm.train()
    pm.freeze_bn()
    om.train()
    for m in om.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=500)
    m1, m2 = O.dropout_masks(n, 1, seed=22)
    lo = S.ce_loss(om(x, (m1[0], m2[0])), lab)
    lo.backward()
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    loss = crit(pm(x.cuda(), dropout_masks=(m1[0].cuda(), m2[0].cuda())), lab.cuda())
    loss.backward()
    assert abs(loss.item() - lo.item()) <= 1e-5 * abs(lo.item())
    # Self-calibrating bound: an f64 run of the oracle is the truth; the HIP gradients may be no further from it
    # than a small multiple of what stock f32 PyTorch itself is (ReLU/ReLU6 kinks make this net's gradients
    # sensitive to f32 rounding: the f32-vs-f64 distance below is the noise floor, measured in the same run).
    o64 = O.ODeepLab(backbone, 16, ncls)
    O.fill_state_dict(o64, seed=21)
    o64 = o64.double().train()
    for m in o64.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    S.ce_loss(o64(x.double(), (m1[0].double(), m2[0].double())), lab).backward()
    g64 = {k: p.grad for k, p in o64.named_parameters()}
    g32 = {k: p.grad.double() for k, p in om.named_parameters()}
    floor = 1e-3 * float(np.median([v.norm().item() for v in g64.values()]))
    rel = lambda g, k: (g - g64[k]).norm().item() / max(g64[k].norm().item(), floor)  # noqa: E731
    hip = sorted(((rel(p.grad.double().cpu(), k), k) for k, p in pm.named_parameters()), reverse=True)
    cpu = sorted(((rel(g32[k], k), k) for k in g64), reverse=True)
    med_hip, med_cpu = float(np.median([r for r, _ in hip])), float(np.median([r for r, _ in cpu]))
    print("%s frozen-BN grads vs f64 oracle: HIP worst %.2e median %.2e | stock f32 CPU worst %.2e median %.2e"
          % (backbone, hip[0][0], med_hip, cpu[0][0], med_cpu))
    if backbone == "resnet":
        # ReLU-only net: the typical (median) parameter agrees at the f32 rounding level (measured 3e-6 vs 2e-6).  The
        # WORST parameter is rounding luck: one ReLU input within rounding of 0 flips its gate and moves the gradients
        # of the few layers upstream by ~1e-3 -- tests/dev/grad_modes.py shows it hitting the f32-MFMA, the bf16x6 and
        # the stock-PyTorch path alike depending on the weight seed -- so it is bounded at the flip scale.
        assert med_hip <= max(4 * med_cpu, 1e-5) and hip[0][0] <= 5e-3, (hip[:3], cpu[:3])
    else:
        # MobileNetV2 at random init saturates 5-15 % of its ReLU6 units: an activation that sits within rounding
        # of 0 or 6 flips its gradient gate, and ONE flip on the 5x5 maps moves every upstream gradient by ~1e-2.
        # Forward errors of HIP and stock f32 vs f64 are equal layer by layer (tests/dev/debug_fwd.py) and every
        # InvertedResidual block is exact in isolation (test_inverted_residual_blocks_exact below); which kinks
        # flip is rounding luck, so the whole-net bound is the flip scale, not the rounding scale.
        assert med_hip <= 2e-2 and hip[0][0] <= 5e-2, (hip[:3], cpu[:3])


def test_inverted_residual_blocks_exact():
    """every MobileNetV2 block shape (expand 1x1 over the fixed_padding border, depthwise 3x3 s1/s2/dilated,
    linear 1x1, residual) forward + all gradients, eval- and train-mode BN, against an f64 oracle: the HIP
    path must be as close to f64 as stock f32 PyTorch is."""
    ops, O, S = _setup()
    import torch.nn as nn
    from models.backbone.mobilenet import InvertedResidual

    for (cin, cout, stride, dil, t, hw) in [(160, 160, 1, 1, 6, 5), (96, 160, 2, 1, 6, 9), (160, 320, 1, 2, 6, 5),
                                            (32, 16, 1, 1, 1, 33), (24, 24, 1, 1, 6, 17)]:
        ob = O.OInvertedResidual(cin, cout, stride, dil, t)
        O.fill_state_dict(ob, seed=3)
        pb = InvertedResidual(cin, cout, stride, dil, t, nn.BatchNorm2d)
        pb.load_state_dict(ob.state_dict())
        pb = pb.cuda()
        for train in (False, True):
            ob.train(train)
            pb.train(train)
            o64 = O.OInvertedResidual(cin, cout, stride, dil, t)
            o64.load_state_dict(ob.state_dict())
            o64 = o64.double().train(train)
            x = torch.randn(2, cin, hw, hw, generator=torch.Generator().manual_seed(cin + hw))
            res = {}
            for tag, mod, xx in (("f64", o64, x.double()), ("f32", ob, x.clone()), ("hip", pb, x.cuda())):
                mod.zero_grad()
                xx = xx.requires_grad_(True)
                y = mod(xx)
                go = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).to(y.dtype).to(y.device)
                y.backward(go)
                res[tag] = dict(y=y.detach().double().cpu(), dx=xx.grad.double().cpu(),
                                **{k: p.grad.double().cpu() for k, p in mod.named_parameters()})
            for k, ref in res["f64"].items():
                e32 = (res["f32"][k] - ref).norm().item() / max(ref.norm().item(), 1e-12)
                ehip = (res["hip"][k] - ref).norm().item() / max(ref.norm().item(), 1e-12)
                assert ehip <= 5 * e32 + 2e-6, ((cin, cout, stride, dil, t, hw), train, k, e32, ehip)


def test_region_maps_and_remaining_ceal_selectors_vs_oracle():
    """create_region_maps (mc_dropout.py:123-171) end to end on a synthetic pool, and the margin / fusion
    selectors (ceal.py:72-98,133-140), against the oracle composing the same steps on the CPU."""
    ops, O, S = _setup()
    import constants
    from active_selection.ceal import ActiveSelectionCEAL
    from active_selection.mc_dropout import ActiveSelectionMCDropout

    ncls, hw, T, region = 19, 65, 4, 17
    pm, om = _product("mobilenet", ncls, O, seed=12)
    pm.eval()
    om.eval()
    keys = [("img_%03d" % i).encode("ascii") for i in range(4)]
    pool = {k: O.synthetic_batch(1, hw, hw, ncls, first_index=700 + i) for i, k in enumerate(keys)}

    def factory(images, include_labels, bs=2):
        for i in range(0, len(images), bs):
            chunk = images[i:i + bs]
            yield {"image": torch.cat([pool[k][0] for k in chunk]), "label": torch.cat([pool[k][1] for k in chunk])}

    # --- CEAL margin + fusion
    sel = ActiveSelectionCEAL(ncls, None, hw, 2, loader_factory=factory)
    with torch.no_grad():
        lo = torch.cat([om(pool[k][0]) for k in keys])
    labs = torch.cat([pool[k][1] for k in keys])
    conf, margin, ent = S.softmax_score_maps(lo, labs, ncls)
    margin_scores = [float(np.mean(m.numpy())) for m in margin]
    assert list(sel.get_least_margin_samples(pm, keys, 2)) == S.select_top(margin_scores, keys, 2, reverse=False)
    fused = sel.get_fusion_of_confidence_margin_entropy_samples(pm, keys, 2)
    assert len(fused) == 2 and set(fused) <= set(keys)

    # --- region maps: votes come from the HIP model (dropout draws are its own), so the oracle re-derives
    # everything downstream of the votes: entropy -> suppression -> box sums -> global min-max -> NMS
    class Recorder(ActiveSelectionMCDropout):
        def _votes(self, model, image_batch, steps, masks=None):
            v = super()._votes(model, image_batch, steps, masks)
            self.seen.append(v.cpu())
            return v

    rsel = Recorder(ncls, None, hw, 2, loader_factory=factory)
    rsel.seen = []
    constants.MC_STEPS = T
    existing = [[], [(5, 5, region, region)], [], []]
    got_regions, got_count = rsel.create_region_maps(pm, keys, existing, region, 2)
    constants.MC_STEPS = 20
    assert all(not m.training for m in pm.modules() if isinstance(m, torch.nn.Dropout2d))  # model.eval() on exit
    votes = torch.cat(rsel.seen)
    assert votes.shape == (4, T, hw, hw) and votes.dtype == torch.uint8
    maps = S.vote_entropy_maps(votes.long(), labs, ncls)
    for i, regs in enumerate(existing):
        S.suppress_labeled(maps[i], regs)
    score = torch.stack([S.box_sum(m, region) for m in maps])
    S.minmax_normalize(score)
    want_regions, want_count = S.square_nms(score, region, (2 * hw * hw) / (region * region))
    assert got_count == want_count
    assert got_regions == {keys[i]: r for i, r in enumerate(want_regions) if r}


@pytest.mark.parametrize("backbone", ["resnet", "mobilenet"])
def test_output_stride_8_vs_oracle(backbone):
    """output_stride = 8 (deeplab.py:16-17, resnet.py:57-63 strides [1,2,1,1] / dilations [1,1,2,4], aspp.py:52-53
    dilations [1,12,24,36], mobilenet.py:99-107): eval-mode logits against the f64 oracle, and one frozen-BN backward --
    dilations 24 and 36 on a 9x9 map put most 3x3 taps in the padding, which the loaders must skip, not read."""
    ops, O, S = _setup()
    from models.deeplab import DeepLab
    from utils.loss import SegmentationLosses

    ncls, n, hw = 19, 2, 65
    om = O.ODeepLab(backbone, 8, ncls)
    O.fill_state_dict(om, seed=33)
    pm = DeepLab(backbone=backbone, output_stride=8, num_classes=ncls, sync_bn=False, freeze_bn=False, pretrained=False)
    assert list(pm.state_dict().keys()) == list(om.state_dict().keys())
    pm.load_state_dict(om.state_dict())
    pm = pm.cuda()
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=900)
    o64 = O.ODeepLab(backbone, 8, ncls)
    O.fill_state_dict(o64, seed=33)
    o64 = o64.double().eval()
    pm.eval()
    with torch.no_grad():
        ref = o64(x.double())
        got = pm(x.cuda()).double().cpu()
    assert got.shape == ref.shape == (n, ncls, hw, hw)
    assert (got - ref).abs().max().item() <= 1e-3 * ref.abs().max().item()
    # frozen-BN backward (running statistics): every gradient against the f64 oracle
    pm.train()
    pm.freeze_bn()
    o64.train()
    for m in o64.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    m1, m2 = O.dropout_masks(n, 1, seed=34)
    crit = SegmentationLosses(cuda=True).build_loss("ce")
    # the HIP forward's own activation gates are replayed inside the f64 oracle (tests/gate_replay.py): both sides then
    # differentiate the same piecewise-linear function and EVERY parameter can be held to the rounding level -- with true
    # ReLUs one pre-activation within rounding of 0 moves all the layers upstream of it by ~1e-3, in any f32 arithmetic
    from gate_replay import GateReplay

    rec = GateReplay(ops)
    try:
        rec.record()
        crit(pm(x.cuda(), dropout_masks=(m1[0].cuda(), m2[0].cuda())), lab.cuda()).backward()
        rec.stop_recording()
        rec.replay()
        S.ce_loss(o64(x.double(), (m1[0].double(), m2[0].double())), lab).backward()
    finally:
        rec.restore()
    assert all(rec.used) and len(rec.gates) >= 40
    g64 = {k: p.grad for k, p in o64.named_parameters()}
    floor = 1e-3 * float(np.median([v.norm().item() for v in g64.values()]))
    rels = sorted((((p.grad.double().cpu() - g64[k]).norm().item() / max(g64[k].norm().item(), floor), k)
                   for k, p in pm.named_parameters()), reverse=True)
    med = float(np.median([r for r, _ in rels]))
    print("os8 %s grads vs f64 oracle (gates injected): worst %.2e (%s) median %.2e over %d parameters, %d gate sites"
          % (backbone, rels[0][0], rels[0][1], med, len(rels), len(rec.gates)))
    assert rels[0][0] <= (5e-5 if backbone == "resnet" else 3e-4) and med <= (1e-5 if backbone == "resnet" else 5e-5), rels[:3]
