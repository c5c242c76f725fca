"""the device-side pool reader.  PathsDataset (this build's dataloaders/dataset/paths_dataset.py over
csrc/pool_reader.hip) on synthetic LMDB-style records against the fixtures written from the REFERENCE's transform classes
(tests/golden/pool_reader.npz, oracle/make_goldens_r2.py): labels, the uint8 resample and the f32 normalisation are equal
bit for bit, for the FixScaleCrop path (landscape, portrait, up-scaling) and the 512 padded-canvas path (crop_size = -1),
with and without labels; then the prefetching loader (order, ragged last batch) and a selector reading its pool through it."""
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases():
    from test_cpu import POOL_CASES, pool_record

    return POOL_CASES, pool_record


def test_paths_dataset_vs_reference_golden():
    from dataloaders.dataset.paths_dataset import DictEnv, PathsDataset

    cases, pool_record = _cases()
    g = np.load(os.path.join(GOLD, "pool_reader.npz"))
    for h, w, crop, seed in cases:
        rec = pool_record(h, w, seed)
        key = ("rec_%d" % seed).encode("ascii")
        env = DictEnv({key: pickle.dumps(rec, protocol=3)})     # the wire format of utils/cityscapes_to_lmdb.py:49
        tag = "%dx%d_c%d" % (h, w, crop)
        sub = slice(None, None, 3) if crop == -1 else slice(None)
        s = PathsDataset(env, [key], crop, include_labels=True)[0]
        size = 512 if crop == -1 else crop
        assert s["image"].shape == (3, size, size) and s["label"].shape == (size, size) and s["image"].is_cuda
        assert np.array_equal(s["label"].cpu().numpy()[sub, sub].astype(np.uint8), g["pool_%s_label" % tag]), tag
        assert np.array_equal(s["image"].cpu().numpy()[:, sub, sub], g["pool_%s_image" % tag]), tag
        only = PathsDataset(env, [key], crop, include_labels=False)[0]
        assert np.array_equal(only.cpu().numpy()[:, sub, sub], g["pool_%s_image_only" % tag]), tag


def test_cityscapes_sized_record_and_loader_order():
    """a full 1024 x 2048 frame (the 2:1 down-scale of the real pool) against the oracle, and the prefetching loader"""
    from dataloaders.dataset.paths_dataset import DictEnv, PathsDataset, pool_loader
    from oracle import transforms_cpu as T

    _, pool_record = _cases()
    rec = pool_record(1024, 2048, 77)
    want = T.pool_sample(rec, 513, True)
    env = DictEnv({b"big": pickle.dumps(rec, protocol=3)})
    got = PathsDataset(env, [b"big"], 513, include_labels=True)[0]
    assert np.array_equal(got["label"].cpu().numpy(), want["label"])
    assert np.array_equal(got["image"].cpu().numpy(), want["image"])
    # loader: 7 records, batch 3 -> 3 + 3 + 1, in key order
    recs = {("k%d" % i).encode(): pickle.dumps(pool_record(60 + i, 90, 100 + i), protocol=3) for i in range(7)}
    keys = sorted(recs)
    batches = list(pool_loader(DictEnv(recs), keys, 33, True, 3, workers=3, ahead=2))
    assert [b["image"].shape[0] for b in batches] == [3, 3, 1]
    flat = torch.cat([b["image"] for b in batches])
    for i, k in enumerate(keys):
        one = PathsDataset(DictEnv(recs), [k], 33, include_labels=True)[0]
        assert torch.equal(flat[i], one["image"]), k
    imgs = list(pool_loader(DictEnv(recs), keys, 33, False, 4))
    assert [b.shape for b in imgs] == [(4, 3, 33, 33), (3, 3, 33, 33)]


def test_selector_reads_its_pool_through_the_device_reader():
    """ActiveSelectionCEAL with NO injected loader: `env` + keys -> this build's PathsDataset / pool_loader; same selection and
    scores as feeding it the oracle-transformed tensors through a loader_factory"""
    from active_selection.ceal import ActiveSelectionCEAL
    from dass_hip import ops
    from dataloaders.dataset.paths_dataset import DictEnv
    from models.deeplab import DeepLab
    from oracle import deeplab_cpu as O
    from oracle import transforms_cpu as T

    _, pool_record = _cases()
    ops.set_compute_dtype(torch.float32)
    om = O.ODeepLab("mobilenet", 16, 19)
    O.fill_state_dict(om, seed=3)
    pm = DeepLab(backbone="mobilenet", output_stride=16, num_classes=19, sync_bn=False, pretrained=False)
    pm.load_state_dict(om.state_dict())
    pm = pm.cuda().eval()
    recs = {("img%d" % i).encode(): pool_record(80 + 3 * i, 140 - 5 * i, 200 + i) for i in range(5)}
    keys = sorted(recs)
    env = DictEnv({k: pickle.dumps(v, protocol=3) for k, v in recs.items()})
    sel_env = ActiveSelectionCEAL(19, env, 65, 2)
    got_sel, got_ent = sel_env.get_maximum_entropy_samples(pm, keys, 2)

    def factory(images, include_labels, bs=2):
        for i in range(0, len(images), bs):
            ss = [T.pool_sample(recs[k], 65, True) for k in images[i:i + bs]]
            yield {"image": torch.from_numpy(np.stack([s["image"] for s in ss])), "label": torch.from_numpy(np.stack([s["label"] for s in ss]))}

    sel_f = ActiveSelectionCEAL(19, None, 65, 2, loader_factory=factory)
    want_sel, want_ent = sel_f.get_maximum_entropy_samples(pm, keys, 2)
    assert list(got_sel) == list(want_sel) and np.array_equal(np.array(got_ent), np.array(want_ent))
