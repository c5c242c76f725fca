"""the noise and max-subset selector families on the HIP kernels."""
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


def test_max_representative_vs_reference_golden():
    ops, O, S = _setup()
    from active_selection import get_max_subset_active_selector

    g = np.load(os.path.join(GOLD, "max_subset.npz"))
    np.random.seed(seed=27)  # the reference's own test inputs (active_selection/tests.py:616-642)
    clusters = [np.random.normal(loc=2.0, scale=1.0, size=(400, 1024)), np.random.normal(loc=4.0, scale=1.0, size=(400, 1024)),
                np.random.normal(loc=6.0, scale=1.0, size=(150, 1024)), np.random.normal(loc=4.0, scale=3.0, size=(50, 1024))]
    images = np.concatenate(clusters, axis=0)
    cands = list(np.random.randint(0, len(images), 8))
    assert cands == g["candidates"].tolist()
    sel = get_max_subset_active_selector(None, None, None)
    got = sel._max_representative_samples(images.astype(np.float32), images[cands, :].astype(np.float32), 4)
    assert got == g["picks"].tolist()
    big = np.asarray(O._hash_uniform(400 * 2736, 123), dtype=np.float32).reshape(400, 2736)
    cidx = list(range(0, 400, 7))
    assert sel._max_representative_samples(big, big[cidx], 20) == g["big_picks"].tolist()


def _pool(O, n, hw, first):
    keys = [("img_%03d" % i).encode("ascii") for i in range(n)]
    pool = {k: O.synthetic_batch(1, hw, hw, 19, first_index=first + i) for i, k in enumerate(keys)}

    def factory(images, include_labels, bs=2):
        for i in range(0, len(images), bs):
            chunk = images[i:i + bs]
            if include_labels:
                yield {"image": torch.cat([pool[k][0] for k in chunk]), "label": torch.cat([pool[k][1] for k in chunk])}
            else:
                yield torch.cat([pool[k][0] for k in chunk])

    return keys, pool, factory


def test_representative_images_vs_oracle():
    ops, O, S = _setup()
    from active_selection.max_subset import ActiveSelectionMaxSubset
    from models.deeplab import DeepLab

    om = O.ODeepLab("mobilenet", 16, 19)
    O.fill_state_dict(om, seed=14)
    pm = DeepLab(backbone="mobilenet", num_classes=19, sync_bn=False, pretrained=False)
    pm.load_state_dict(om.state_dict())
    pm = pm.cuda().eval()
    om.eval()
    keys, pool, factory = _pool(O, 5, 513, 800)
    sel = ActiveSelectionMaxSubset(None, 513, 2, loader_factory=factory)
    got = sel.get_representative_images(pm, keys, keys[1:5])
    om.return_features = True
    with torch.no_grad():
        feats = S.coreset_features(torch.cat([om(pool[k][0])[1] for k in keys]), 64)
    want = S.max_representative_samples(feats, feats[1:5], 2)
    assert got == [keys[1:5][i] for i in want]
    assert pm.return_features is False


def test_noise_selectors_run_and_reduce_like_the_oracle():
    ops, O, S = _setup()
    import constants
    from active_selection import get_active_selection_class
    from models.deeplab import DeepLab

    om = O.ODeepLab("mobilenet", 16, 19)
    O.fill_state_dict(om, seed=15)
    pm = DeepLab(backbone="mobilenet", num_classes=19, sync_bn=False, pretrained=False)
    pm.load_state_dict(om.state_dict())
    pm = pm.cuda().eval()
    keys, pool, factory = _pool(O, 3, 65, 850)
    sel = get_active_selection_class("noise_variance", 19, None, 65, 2)
    sel.loader_factory = factory
    constants.MC_STEPS = 4
    try:
        torch.manual_seed(0)
        x, lab = pool[keys[0]][0].cuda(), pool[keys[0]][1].cuda()
        maps = sel._get_vote_entropy_for_batch_with_input_noise(pm, x, lab)
        assert len(maps) == 1 and maps[0].shape == (65, 65) and float(maps[0].min()) >= 0
        assert float(maps[0][: 65 // 10].abs().max()) == 0.0            # label-masked rows are zeroed
        assert float(maps[0].max()) <= np.log2(4) + 1e-5                 # entropy of 4 votes is at most 2 bits
        assert pm.noisy_features is False
        fmaps = sel._get_vote_entropy_for_batch_with_feature_noise(pm, x, lab)
        assert pm.noisy_features is False and fmaps[0].shape == (65, 65)
        # zero input noise == deterministic votes == zero entropy everywhere
        det = sel._vote_maps(pm, x, lab, perturb=lambda t: t)
        assert float(det[0].abs().max()) == 0.0
        picked = sel.get_vote_entropy_for_images_with_input_noise(pm, keys, 2)
        assert len(picked) == 2 and set(picked) <= set(keys)
        picked = sel.get_vote_entropy_for_batch_with_noise_and_vote_entropy(pm, keys, 1)
        assert len(picked) == 1
        assert all(not m.training for m in pm.modules() if isinstance(m, torch.nn.Dropout2d))
        regions, cnt = sel.create_region_maps(pm, keys, [[], [], []], 17, 1)
        assert cnt >= 1 and all(k in keys for k in regions)
    finally:
        constants.MC_STEPS = 20


def test_evaluator_vs_reference_golden():
    """confusion matrix counted on the device (fused argmax + histogram, or a ready prediction map,
    from device tensors or from the numpy arrays Trainer.validation passes) == the REFERENCE Evaluator's matrix and four
    metrics (tests/golden/metrics.npz, written by oracle/make_goldens_r2.py from utils/metrics.py:6-49)"""
    ops, O, S = _setup()
    from utils.metrics import Evaluator

    gold = np.load(os.path.join(GOLD, "metrics.npz"))
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(3, 19, 33, 41, generator=g)
    target = torch.randint(0, 19, (3, 33, 41), generator=g).float()
    target[:, :4] = 255
    target[0, 5] = -1
    logits2 = torch.randn(3, 19, 33, 41, generator=g)
    dev = Evaluator(19)
    dev.add_batch(target.cuda(), logits.cuda())                                  # fused argmax + histogram
    assert np.array_equal(dev.confusion_matrix, gold["cm1"])
    dev.add_batch(target.numpy(), np.argmax(logits2.numpy(), axis=1))            # what active_train.py:159-163 passes
    assert np.array_equal(dev.confusion_matrix, gold["cm2"])
    names = ("Pixel_Accuracy", "Pixel_Accuracy_Class", "Mean_Intersection_over_Union", "Frequency_Weighted_Intersection_over_Union")
    for fn, want in zip(names, gold["vals"]):
        assert abs(getattr(dev, fn)() - want) < 1e-12, fn
    dev2 = Evaluator(19)
    dev2.add_batch(target.cuda(), logits.argmax(1).cuda())                       # ready prediction map, device tensors
    assert np.array_equal(dev2.confusion_matrix, gold["cm1"])
    dev2.confusion_matrix = gold["cm3"]                                          # an empty class: nan conventions
    for fn, want in zip(names, gold["vals3"]):
        assert abs(getattr(dev2, fn)() - want) < 1e-12, fn
    dev2.reset()
    assert dev2.confusion_matrix.sum() == 0 and dev2.confusion_matrix.shape == (19, 19)
    # the oracle restatement agrees with the same fixtures
    assert np.array_equal(S.confusion_matrix(target.numpy(), np.argmax(logits.numpy(), axis=1), 19), gold["cm1"])


def test_mc_noise_votes_and_entropy_vs_reference_golden():
    """gaussian input noise (mc_noise.py:21-44) and feature noise (mc_noise.py:62-84 over
    deeplab.py:39-56) with numpy's seeded stream replayed through the product's noise hooks: argmax votes equal the
    reference's wherever its top-2 margin exceeds 1e-3, and the entropy maps follow."""
    ops, O, S = _setup()
    import constants
    from active_selection.mc_noise import ActiveSelectionMCNoise
    from models.deeplab import DeepLab

    gold = np.load(os.path.join(GOLD, "mc_noise.npz"))
    n, hw, ncls, T = [int(v) for v in gold["meta"]]
    om = O.ODeepLab("mobilenet", 16, ncls)
    O.fill_state_dict(om, seed=15)
    pm = DeepLab(backbone="mobilenet", output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False)
    pm.load_state_dict(om.state_dict())
    pm = pm.cuda().eval()
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=60)

    def np_draw(shape, sigma):  # the reference's draw (mc_noise.py:24, deeplab.py:40)
        return torch.from_numpy(np.random.normal(loc=0.0, scale=sigma, size=shape).astype(np.float32))

    class Recorder(ActiveSelectionMCNoise):
        def _vote_maps(self, model, image_batch, label_batch, perturb=None):
            seen = []

            def spy(inp):
                out = model(inp)
                seen.append(torch.argmax(out, dim=1).cpu())
                return out

            maps = super()._vote_maps(spy, image_batch, label_batch, perturb)
            self.votes = torch.stack(seen, 1)
            return maps

    sel = Recorder(ncls, None, hw, n, noise_source=np_draw)
    constants.MC_STEPS = T
    try:
        for tag, seed, fn in (("input", 501, sel._get_vote_entropy_for_batch_with_input_noise),
                              ("feature", 502, sel._get_vote_entropy_for_batch_with_feature_noise)):
            np.random.seed(seed)
            maps = torch.stack(fn(pm, x.cuda(), lab.cuda())).cpu()
            ref_votes = torch.from_numpy(gold[tag + "_votes"]).long()
            safe = torch.from_numpy(gold[tag + "_margin"].astype(np.float32)) > 1e-3
            assert torch.equal(sel.votes[safe], ref_votes[safe]), tag
            flips = int((sel.votes != ref_votes).sum())
            assert flips <= int((~safe).sum())
            if flips == 0:
                assert (maps - torch.from_numpy(gold[tag + "_entropy"])).abs().max().item() <= 1e-5, tag
            # the reduction kernel on the reference's own votes -> the reference's entropy maps
            emap, _ = ops.vote_entropy(torch.from_numpy(gold[tag + "_votes"]).cuda(), lab.cuda(), ncls)
            assert (emap.cpu() - torch.from_numpy(gold[tag + "_entropy"])).abs().max().item() <= 1e-5
        assert pm.noisy_features is False and pm.noise_source is None
    finally:
        constants.MC_STEPS = 20


def test_representative_regions_and_updated_distances_vs_oracle():
    """max_subset.py:115-128 (region branch of `variance_representative`) and core_set.py:32-38 through the product
    surface, against the oracle composing the same steps on the CPU model"""
    ops, O, S = _setup()
    from active_selection.core_set import ActiveSelectionCoreSet
    from active_selection.max_subset import ActiveSelectionMaxSubset
    from models.deeplab import DeepLab

    ncls, hw, region = 19, 129, 33
    om = O.ODeepLab("mobilenet", 16, ncls)
    O.fill_state_dict(om, seed=17)
    pm = DeepLab(backbone="mobilenet", output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False)
    pm.load_state_dict(om.state_dict())
    pm = pm.cuda().eval()
    om.eval()
    keys, pool, factory = _pool(O, 4, hw, 900)

    class Wrapper(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.module = m

        def forward(self, x):
            return self.module(x)

    sel = ActiveSelectionMaxSubset(None, hw, 2, loader_factory=factory)
    candidates = {keys[0]: [(0, 0, region, region), (40, 64, region, region)], keys[2]: [(96, 96, region, region)],
                  keys[3]: [(10, 50, region, region), (64, 0, region, region), (90, 20, region, region)]}
    got_regions, got_count = sel.get_representative_regions(Wrapper(pm), keys, candidates, region)
    om.return_features = True
    with torch.no_grad():
        feats = {k: om(pool[k][0])[1] for k in keys}
    grid = np.concatenate([S.region_grid_features(feats[k], region, hw) for k in keys])
    li, lr = sel._convert_regions_to_list(candidates)
    cand = np.concatenate([S.region_features(feats[k], [r], hw) for k, r in zip(li, lr)])
    got_grid = sel._get_features_for_image_regions(Wrapper(pm), keys, region).cpu().numpy()
    got_cand = sel._get_features_for_regions(Wrapper(pm), li, lr).cpu().numpy()
    assert got_grid.shape == grid.shape and np.abs(got_grid - grid).max() <= 1e-3
    assert got_cand.shape == cand.shape == (6, 304) and np.abs(got_cand - cand).max() <= 1e-3
    picks = S.max_representative_samples(grid, cand, len(cand) // 2)
    want = {}
    for i in picks:
        want.setdefault(li[i], []).append(lr[i])
    assert got_count == len(picks) == 3 and got_regions == want
    assert pm.return_features is False
    # _updated_distances: the reference's two call forms (core_set.py:19,25)
    cs = ActiveSelectionCoreSet(None, None, None)
    f = np.asarray(O._hash_uniform(50 * 64, 5), dtype=np.float32).reshape(50, 64)
    from sklearn.metrics import pairwise_distances

    d0 = cs._updated_distances([3, 7, 11], f, None)
    want0 = np.min(pairwise_distances(f.astype(np.float64), f[[3, 7, 11]].astype(np.float64)), axis=1).reshape(-1, 1)
    assert d0.shape == (50, 1) and d0.dtype == np.float64 and np.abs(d0 - want0).max() <= 1e-12
    d1 = cs._updated_distances([20], f, d0)
    want1 = np.minimum(want0, pairwise_distances(f.astype(np.float64), f[[20]].astype(np.float64)))
    assert np.abs(d1 - want1).max() <= 1e-12


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_overfit_one_batch_loss_goes_down(dtype):
    """plumbing check of the whole training path (config 0 of BASELINE.json in spirit): SGD on one fixed synthetic
    batch must drive the loss down, in both numerics modes"""
    ops, O, S = _setup()
    from models.deeplab import DeepLab
    from utils.loss import SegmentationLosses

    ops.set_compute_dtype(torch.float32 if dtype == "f32" else torch.bfloat16)
    try:
        torch.manual_seed(0)
        model = DeepLab(backbone="mobilenet", num_classes=4, sync_bn=False, pretrained=False).cuda().train()
        x = torch.randn(2, 3, 65, 65, generator=torch.Generator().manual_seed(1)).cuda()
        y = torch.zeros(2, 65, 65)
        y[:, 32:, :32], y[:, :32, 32:], y[:, 32:, 32:] = 1, 2, 3          # quadrant labels: learnable from position cues
        y[:, :5] = 255
        y = y.cuda()
        crit = SegmentationLosses(cuda=True).build_loss("ce")
        opt = torch.optim.SGD([{"params": model.get_1x_lr_params(), "lr": 0.01}, {"params": model.get_10x_lr_params(), "lr": 0.1}],
                              momentum=0.9, weight_decay=5e-4)
        losses = []
        for _ in range(40):
            opt.zero_grad()
            loss = crit(model(x), y)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        print(dtype, "loss %.4f -> %.4f" % (losses[0], losses[-1]))
        assert all(np.isfinite(losses)) and losses[-1] < 0.5 * losses[0]
    finally:
        ops.set_compute_dtype(torch.float32)
