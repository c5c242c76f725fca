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
    with pytest.raises(NotImplementedError):
        sel.get_representative_regions(None, [], {}, 129)


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


def test_evaluator_device_confusion_matrix_vs_reference_formulas():
    """argmax + confusion matrix on the device == the reference's numpy argmax + bincount"""
    ops, O, S = _setup()
    from utils.metrics import Evaluator

    g = torch.Generator().manual_seed(3)
    logits = torch.randn(3, 19, 33, 41, generator=g)
    target = torch.randint(0, 19, (3, 33, 41), generator=g).float()
    target[:, :4] = 255
    target[0, 5] = -1
    ref = Evaluator(19)  # numpy path == utils/metrics.py:37-46 line for line
    ref.add_batch(target.numpy(), np.argmax(logits.numpy(), axis=1))
    dev = Evaluator(19)
    dev.add_batch(target.cuda(), logits.cuda())                       # fused argmax + histogram
    assert np.array_equal(dev.confusion_matrix, ref.confusion_matrix)
    dev2 = Evaluator(19)
    dev2.add_batch(target.cuda(), logits.argmax(1).cuda())            # ready prediction map
    dev2.add_batch(target.cuda(), logits.argmax(1).cuda())
    assert np.array_equal(dev2.confusion_matrix, 2 * ref.confusion_matrix)
    for fn in ("Pixel_Accuracy", "Pixel_Accuracy_Class", "Mean_Intersection_over_Union",
               "Frequency_Weighted_Intersection_over_Union"):
        assert abs(getattr(dev, fn)() - getattr(ref, fn)()) < 1e-12
    dev.reset()
    assert dev.confusion_matrix.sum() == 0


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
