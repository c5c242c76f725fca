"""Round-2 supplement of oracle/make_goldens.py: pins the oracle restatements added this round against the reference
and writes the extra fixtures (existing fixtures are left untouched).

TEST INFRASTRUCTURE.  Run ONLY in the authoring container (needs /root/reference, read-only):
    python oracle/make_goldens_r2.py
  metrics.npz   utils/metrics.py:6-49 Evaluator on seeded label / prediction maps: confusion matrix + the four metrics
  pool_reader.npz  dataloaders/dataset/paths_dataset.py:27-52 over the reference's own transform classes
                (custom_transforms.py FixScaleCrop / ScaleWithPadding / Normalize / ToTensor) on synthetic LMDB-style records;
                scipy.misc.imresize (removed from SciPy) is supplied as its published two-line body over PIL, torchvision's
                ToTensor / Normalize (absent here) by their documented float32 arithmetic -- see pool_reader_goldens()
  mc_noise.npz  active_selection/mc_noise.py:21-44 (gaussian input noise) and :62-84 (+ models/deeplab.py:39-56 feature
                noise) run on the reference DeepLab-MobileNet with numpy's seeded generator: the reference's argmax votes
                (recorded by a wrapper around the model) and its entropy maps.  The tests replay the same np.random
                stream through the product's noise hooks.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import deeplab_cpu as O  # noqa: E402
from oracle import selection_cpu as S  # noqa: E402
from oracle.make_goldens import OUT, import_reference, maxdiff  # noqa: E402


def np_draw(shape, scale):
    """the reference's draw: np.random.normal(loc=0.0, scale=scale, size=shape).astype(np.float32)"""
    return torch.from_numpy(np.random.normal(loc=0.0, scale=scale, size=shape).astype(np.float32))


class Recorder(torch.nn.Module):
    """DataParallel-style wrapper (`.module`) that records the argmax of every forward"""

    def __init__(self, module):
        super().__init__()
        self.module = module
        self.votes = []

    def forward(self, x):
        out = self.module(x)
        self.votes.append(torch.argmax(out, dim=1))
        return out


def pool_record(h, w, seed):
    """synthetic LMDB-style record: smooth-ish RGB + a blocky label plane with some 255 (ignore) pixels, uint8 [h, w, 4]"""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([(127 + 100 * np.sin(yy / (3.0 + c) + xx / (5.0 - c)) + rng.randint(-20, 21, (h, w))).clip(0, 255) for c in range(3)], 2)
    lab = ((yy // 7 + xx // 11) % 19).astype(np.uint8)
    lab[rng.rand(h, w) < 0.03] = 255
    return np.ascontiguousarray(np.dstack((rgb.astype(np.uint8), lab)))


POOL_CASES = [(96, 192, 65, 1), (130, 100, 65, 2), (64, 64, 65, 3), (75, 100, 129, 4), (96, 192, -1, 5), (150, 101, -1, 6)]


def pool_reader_goldens(out):
    """drives the reference's transform classes the way paths_dataset.py:40-52 composes them"""
    import scipy.misc
    from PIL import Image

    def imresize(arr, size, interp="bilinear", mode=None):  # scipy.misc.imresize (SciPy <= 1.2) for uint8 input and a size tuple
        im = Image.fromarray(arr, mode=mode)
        func = {"nearest": 0, "lanczos": 1, "bilinear": 2, "bicubic": 3, "cubic": 3}
        return np.asarray(im.resize((size[1], size[0]), resample=func[interp]))

    scipy.misc.imresize = imresize
    from dataloaders import custom_transforms as rtr

    rtr.imresize = imresize  # the module did `from scipy.misc import imresize` at import time
    from oracle import transforms_cpu as T

    for h, w, crop, seed in POOL_CASES:
        rec = pool_record(h, w, seed)
        image, target = rec[:, :, 0:3], rec[:, :, 3]
        tag = "%dx%d_c%d" % (h, w, crop)
        # include_labels=True: scalecrop -> custom Normalize -> custom ToTensor (paths_dataset.py:40-45)
        scalecrop = rtr.ScaleWithPadding(base_size=512) if crop == -1 else rtr.FixScaleCrop(crop_size=crop)
        sample = rtr.ToTensor()(rtr.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])(scalecrop({"image": image, "label": target})))
        mine = T.pool_sample(rec, crop, True)
        assert np.array_equal(sample["label"].numpy(), mine["label"]), tag
        assert np.array_equal(sample["image"].numpy(), mine["image"]), (tag, np.abs(sample["image"].numpy() - mine["image"]).max())
        # include_labels=False: scalecrop_image_only (reference class) -> torchvision ToTensor + Normalize (restated: f32)
        only = rtr.ScaleWithPaddingImageOnly(base_size=512) if crop == -1 else rtr.FixScaleCropImageOnly(crop_size=crop)
        arr = only(image)
        t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
        t = t.float().div(255) if arr.dtype == np.uint8 else t.float()
        mean, std = torch.tensor([0.485, 0.456, 0.406])[:, None, None], torch.tensor([0.229, 0.224, 0.225])[:, None, None]
        t = t.sub(mean).div(std)
        mine_img = T.pool_sample(rec, crop, False)
        assert np.array_equal(t.numpy(), mine_img), tag
        sub = slice(None, None, 3) if crop == -1 else slice(None)   # the 512 canvas is stored subsampled
        out["pool_%s_image" % tag] = sample["image"].numpy()[:, sub, sub]
        out["pool_%s_label" % tag] = sample["label"].numpy()[sub, sub].astype(np.uint8)
        out["pool_%s_image_only" % tag] = t.numpy()[:, sub, sub]
    return out


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = import_reference()
    torch.Tensor.cuda = lambda self, *a, **k: self  # the reference calls .cuda() on host tensors (mc_noise.py:25, deeplab.py:41)

    # ---------------------------------------------------------------- utils/metrics.py
    from utils.metrics import Evaluator

    g = torch.Generator().manual_seed(3)
    logits = torch.randn(3, 19, 33, 41, generator=g)
    target = torch.randint(0, 19, (3, 33, 41), generator=g).float()
    target[:, :4] = 255
    target[0, 5] = -1
    pred = np.argmax(logits.numpy(), axis=1)            # active_train.py:159-163
    ev = Evaluator(19)
    ev.add_batch(target.numpy(), pred)
    cm1 = ev.confusion_matrix.copy()
    assert np.array_equal(cm1, S.confusion_matrix(target.numpy(), pred, 19))
    logits2 = torch.randn(3, 19, 33, 41, generator=g)
    pred2 = np.argmax(logits2.numpy(), axis=1)
    ev.add_batch(target.numpy(), pred2)
    cm2 = ev.confusion_matrix.copy()
    vals = dict(pixel_acc=ev.Pixel_Accuracy(), class_acc=ev.Pixel_Accuracy_Class(), miou=ev.Mean_Intersection_over_Union(),
                fwiou=ev.Frequency_Weighted_Intersection_over_Union())
    mine = S.confusion_metrics(cm2)
    assert all(abs(vals[k] - mine[k]) < 1e-15 for k in vals), (vals, mine)
    # a matrix with an empty class row (nan conventions)
    cm3 = cm2.copy()
    cm3[7, :] = 0
    cm3[:, 7] = 0
    ev.confusion_matrix = cm3
    vals3 = dict(pixel_acc=ev.Pixel_Accuracy(), class_acc=ev.Pixel_Accuracy_Class(), miou=ev.Mean_Intersection_over_Union(),
                 fwiou=ev.Frequency_Weighted_Intersection_over_Union())
    mine3 = S.confusion_metrics(cm3)
    assert all(abs(vals3[k] - mine3[k]) < 1e-15 for k in vals3)
    np.savez_compressed(os.path.join(OUT, "metrics.npz"), cm1=cm1, cm2=cm2, cm3=cm3,
                        vals=np.array([vals[k] for k in ("pixel_acc", "class_acc", "miou", "fwiou")]),
                        vals3=np.array([vals3[k] for k in ("pixel_acc", "class_acc", "miou", "fwiou")]))

    # ---------------------------------------------------------------- active_selection/mc_noise.py
    from active_selection.mc_noise import ActiveSelectionMCNoise

    ncls, n, hw, T = 19, 2, 65, 4
    rm = ref["DeepLab"](backbone="mobilenet", output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False)
    om = O.ODeepLab("mobilenet", 16, ncls)
    O.fill_state_dict(om, seed=15)
    rm.load_state_dict(om.state_dict())
    rm.eval()
    om.eval()
    x, lab = O.synthetic_batch(n, hw, hw, ncls, first_index=60)
    ref["constants"].MC_STEPS = T
    sel = ActiveSelectionMCNoise(ncls, None, hw, n)
    out = {}
    # (a) gaussian input noise, sigma 0.125 (mc_noise.py:21-44)
    rec = Recorder(rm)
    np.random.seed(501)
    ent_ref = torch.stack(sel._get_vote_entropy_for_batch_with_input_noise(rec, x, lab))
    votes_ref = torch.stack(rec.votes, 1)
    np.random.seed(501)
    with torch.no_grad():
        votes_or = torch.stack([torch.argmax(om(x + np_draw(tuple(x.shape), 0.125)), dim=1) for _ in range(T)], 1)
    assert int((votes_or != votes_ref).sum()) == 0
    assert maxdiff(torch.stack(S.vote_entropy_maps(votes_ref, lab, ncls)), ent_ref) == 0.0
    out["input_votes"], out["input_entropy"] = votes_ref.numpy().astype(np.uint8), ent_ref.numpy()
    # (b) feature noise (mc_noise.py:62-84 driving deeplab.py:39-56)
    rec = Recorder(rm)
    np.random.seed(502)
    ent_ref = torch.stack(sel._get_vote_entropy_for_batch_with_feature_noise(rec, x, lab))
    votes_ref = torch.stack(rec.votes, 1)
    assert rm.noisy_features is False
    np.random.seed(502)
    with torch.no_grad():
        votes_or = torch.stack([torch.argmax(om(x, noise=np_draw), dim=1) for _ in range(T)], 1)
    flips = int((votes_or != votes_ref).sum())
    assert flips == 0, flips
    assert maxdiff(torch.stack(S.vote_entropy_maps(votes_ref, lab, ncls)), ent_ref) == 0.0
    out["feature_votes"], out["feature_entropy"] = votes_ref.numpy().astype(np.uint8), ent_ref.numpy()
    # margins of the reference passes (where exactness of the product's votes is demanded), recomputed with the oracle
    for tag, seed in (("input", 501), ("feature", 502)):
        np.random.seed(seed)
        with torch.no_grad():
            tops = []
            for _ in range(T):
                lo = om(x + np_draw(tuple(x.shape), 0.125)) if tag == "input" else om(x, noise=np_draw)
                top = lo.topk(2, dim=1)[0]
                tops.append(top[:, 0] - top[:, 1])
        out[tag + "_margin"] = torch.stack(tops, 1).numpy().astype(np.float16)
    out["meta"] = np.array([n, hw, ncls, T])
    np.savez_compressed(os.path.join(OUT, "mc_noise.npz"), **out)
    ref["constants"].MC_STEPS = 20
    np.savez_compressed(os.path.join(OUT, "pool_reader.npz"), **pool_reader_goldens({}))
    print("metrics.npz, mc_noise.npz, pool_reader.npz written; oracle == reference")


if __name__ == "__main__":
    main()
