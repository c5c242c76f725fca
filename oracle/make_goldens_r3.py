"""Round-3 supplement of oracle/make_goldens.py: EXECUTES the reference's public selector methods (not a restatement of
them) and the reference U-Net of BASELINE config 0, and stores what they return.

TEST INFRASTRUCTURE.  Run ONLY in the authoring container (needs /root/reference, read-only):
    python oracle/make_goldens_r3.py
  selectors_ref.npz   active_selection/ceal.py:19-166 (get_least_confident_samples, get_least_margin_samples,
                      get_maximum_entropy_samples, get_weakly_labeled_data), active_selection/mc_dropout.py:173-196
                      (get_vote_entropy_for_images) and active_selection/core_set.py:40-69 (get_k_center_greedy_selections)
                      run AS WRITTEN on the reference DeepLab-MobileNet.  What the harness supplies, none of it arithmetic:
                        * `Tensor.cuda` is the identity and torch.cuda.FloatTensor the CPU type (no GPU here; as in r2),
                        * `np.bool` = bool (the alias numpy removed in 1.24; ceal.py:84,160 use it as a dtype),
                        * `dataloaders.dataset.paths_dataset.PathsDataset` is replaced in that module's namespace by a
                          Dataset that serves seeded synthetic samples by key (no LMDB in the image) -- the selectors'
                          own torch DataLoader, batching, loops, reductions, sorts and sklearn calls all run unchanged,
                        * for the MC-dropout method the two nn.Dropout2d modules are replaced by mask multipliers fed from
                          oracle.deeplab_cpu.dropout_masks per forward call, so that the product can replay the same masks.
  unet_config0.npz    models/unet.py:18-71 UNet(3, 4) on the config-0 batch (128 x 128, batch 2): eval logits checksum rows
                      and the 3-step SGD loss trajectory; asserts oracle/unet_cpu.py reproduces both bit for bit.
Fixtures are data (selections, scores, label maps, loss values); no reference source is copied.
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
from oracle import unet_cpu as U  # noqa: E402
from oracle.make_goldens import OUT, MaskDropout, import_reference  # noqa: E402

CEAL = dict(ncls=19, n=10, hw=65, batch=4, first_index=300, seed=51)
MCD = dict(ncls=19, n=6, hw=65, batch=4, first_index=340, seed=52, T=4, mask_seed=77)
CORE = dict(ncls=19, n_sel=3, n_cand=7, hw=513, batch=4, first_index=380, seed=53, k=3)


def keys_for(cfg, n):
    return [("img_%04d" % (cfg["first_index"] + i)).encode("ascii") for i in range(n)]


class SyntheticPaths(torch.utils.data.Dataset):
    """stands in for PathsDataset(env, paths, crop_size, include_labels): seeded synthetic samples looked up by key"""
    POOL = {}

    def __init__(self, env, paths, crop_size, include_labels=False):
        self.paths, self.include_labels = paths, include_labels

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, index):
        image, label = SyntheticPaths.POOL[self.paths[index]]
        return {"image": image, "label": label} if self.include_labels else image


def fill_pool(cfg, n):
    x, lab = O.synthetic_batch(n, cfg["hw"], cfg["hw"], cfg["ncls"], first_index=cfg["first_index"])
    keys = keys_for(cfg, n)
    for i, k in enumerate(keys):
        SyntheticPaths.POOL[k] = (x[i], lab[i])
    return keys, x, lab


class Wrapped(torch.nn.Module):
    """DataParallel-style holder (`.module`, core_set.py:44,52); optionally feeds per-call dropout masks"""

    def __init__(self, module, masks=None, steps=None, batch=None):
        super().__init__()
        self.module = module
        self.masks, self.steps, self.batch, self.calls = masks, steps, batch, 0

    def forward(self, x):
        if self.masks is not None:
            b, t = divmod(self.calls, self.steps)       # the selector runs `steps` forwards per loader batch
            rows = slice(b * self.batch, b * self.batch + x.shape[0])
            self.module.aspp.dropout.mask = self.masks[0][t, rows]
            self.module.decoder.last_conv[6].mask = self.masks[1][t, rows]
            self.calls += 1
        return self.module(x)


def ref_mobilenet(ref, seed, ncls):
    rm = ref["DeepLab"](backbone="mobilenet", output_stride=16, num_classes=ncls, sync_bn=False, pretrained=False)
    om = O.ODeepLab("mobilenet", 16, ncls)
    O.fill_state_dict(om, seed=seed)
    rm.load_state_dict(om.state_dict())
    return rm.eval(), om.eval()


def order_of(keys, selected):
    return np.array([keys.index(k) for k in selected], dtype=np.int64)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = import_reference()
    torch.Tensor.cuda = lambda self, *a, **k: self
    if not hasattr(np, "bool"):
        np.bool = bool
    from dataloaders.dataset import paths_dataset as ref_paths

    ref_paths.PathsDataset = SyntheticPaths
    from active_selection.ceal import ActiveSelectionCEAL
    from active_selection.core_set import ActiveSelectionCoreSet
    from active_selection.mc_dropout import ActiveSelectionMCDropout

    out = {}
    # ---------------------------------------------------------------- ceal.py:19-166, executed
    c = CEAL
    keys, x, lab = fill_pool(c, c["n"])
    rm, om = ref_mobilenet(ref, c["seed"], c["ncls"])
    sel = ActiveSelectionCEAL(c["ncls"], None, c["hw"], c["batch"])
    conf_order = order_of(keys, sel.get_least_confident_samples(rm, keys, c["n"]))
    margin_order = order_of(keys, sel.get_least_margin_samples(rm, keys, c["n"]))
    ent_sel, entropies = sel.get_maximum_entropy_samples(rm, keys, c["n"])
    ent_order = order_of(keys, ent_sel)
    thr = float(np.median(entropies))
    weak = sel.get_weakly_labeled_data(rm, keys, thr, entropies=list(entropies))
    weak_keys = list(weak.keys())
    assert weak_keys == [k for k, e in zip(keys, entropies) if e < thr] and 0 < len(weak_keys) < c["n"]
    # the oracle's restatement gives the same orders and scores (this is what pins oracle/selection_cpu.py for a11)
    with torch.no_grad():
        logits = om(x)
    conf, margin, ent = S.softmax_score_maps(logits, lab, c["ncls"])
    o_conf, o_margin, o_ent = (m.mean(dim=(1, 2)).numpy() for m in (conf, margin, ent))
    assert np.array_equal(order_of(keys, S.select_top(o_conf.tolist(), keys, c["n"], reverse=False)), conf_order)
    assert np.array_equal(order_of(keys, S.select_top(o_margin.tolist(), keys, c["n"], reverse=False)), margin_order)
    assert np.array_equal(order_of(keys, S.select_top(o_ent.tolist(), keys, c["n"], reverse=True)), ent_order)
    assert np.abs(np.asarray(entropies, dtype=np.float64) - o_ent).max() < 1e-6
    wl = S.weak_label_maps(logits, lab, c["ncls"])
    for k in weak_keys:
        assert np.array_equal(weak[k], wl[keys.index(k)])
    top = torch.softmax(logits, 1).topk(2, dim=1)[0]
    out.update(ceal_conf_order=conf_order, ceal_margin_order=margin_order, ceal_entropy_order=ent_order,
               ceal_entropies=np.asarray(entropies, dtype=np.float64), ceal_conf_scores=o_conf, ceal_margin_scores=o_margin,
               ceal_threshold=np.float64(thr), ceal_weak_index=order_of(keys, weak_keys),
               ceal_weak_labels=np.stack([weak[k] for k in weak_keys]),
               ceal_logit_margin=(logits.topk(2, dim=1)[0][:, 0] - logits.topk(2, dim=1)[0][:, 1]).numpy().astype(np.float16),
               ceal_prob_margin=(top[:, 0] - top[:, 1]).numpy().astype(np.float16))

    # ---------------------------------------------------------------- mc_dropout.py:173-196, executed
    c = MCD
    keys, x, lab = fill_pool(c, c["n"])
    rm, om = ref_mobilenet(ref, c["seed"], c["ncls"])
    rm.aspp.dropout = MaskDropout()
    rm.decoder.last_conv[6] = MaskDropout()
    m1, m2 = O.dropout_masks(c["n"], c["T"], seed=c["mask_seed"])     # [T, N_pool, 256]: row = position in the pool
    ref["constants"].MC_STEPS = c["T"]
    sel = ActiveSelectionMCDropout(c["ncls"], None, c["hw"], c["batch"])
    wrapped = Wrapped(rm, masks=(m1, m2), steps=c["T"], batch=c["batch"])
    with torch.no_grad():
        mc_order = order_of(keys, sel.get_vote_entropy_for_images(wrapped, keys, c["n"]))
    assert wrapped.calls == c["T"] * ((c["n"] + c["batch"] - 1) // c["batch"])
    votes = S.mc_votes(om, x, (m1, m2))
    o_scores = np.array([float(e.mean()) for e in S.vote_entropy_maps(votes, lab, c["ncls"])])
    assert np.array_equal(order_of(keys, S.select_top(o_scores.tolist(), keys, c["n"], reverse=True)), mc_order)
    ref["constants"].MC_STEPS = 20
    out.update(mc_order=mc_order, mc_scores=o_scores, mc_votes=votes.numpy().astype(np.uint8))

    # ---------------------------------------------------------------- core_set.py:40-69, executed (513^2: FEATURE_DIM 2736)
    c = CORE
    keys, x, lab = fill_pool(c, c["n_sel"] + c["n_cand"])
    rm, om = ref_mobilenet(ref, c["seed"], c["ncls"])
    sel = ActiveSelectionCoreSet(None, c["hw"], c["batch"])
    picked = sel.get_k_center_greedy_selections(c["k"], Wrapped(rm), keys[c["n_sel"]:], keys[:c["n_sel"]])
    core_picks = order_of(keys, picked)
    assert rm.return_features is False
    om.return_features = True
    with torch.no_grad():
        feats = np.concatenate([S.coreset_features(om(x[i:i + c["batch"]])[1]) for i in range(0, x.shape[0], c["batch"])])
    o_picks, _ = S.kcenter_greedy(feats, list(range(c["n_sel"])), c["k"])
    assert list(core_picks) == o_picks, (core_picks, o_picks)
    out.update(core_picks=core_picks, core_features=feats.astype(np.float32)[:, ::16])
    np.savez_compressed(os.path.join(OUT, "selectors_ref.npz"), **out)

    # ---------------------------------------------------------------- config 0: models/unet.py
    from models.unet import UNet

    torch.manual_seed(1234)
    ru = UNet(3, 4)
    ou = U.OUNet(3, 4)
    assert list(ru.state_dict().keys()) == list(ou.state_dict().keys())
    ou.load_state_dict(ru.state_dict())
    xb, yb = U.config0_batch()
    ru.eval(), ou.eval()
    with torch.no_grad():
        r_logits, o_logits = ru(xb), ou(xb)
    assert float((r_logits - o_logits).abs().max()) == 0.0
    crit = ref["SegmentationLosses"](cuda=False).build_loss(mode="ce")
    opt = torch.optim.SGD(ru.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4, nesterov=False)
    ru.train()
    r_losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = crit(ru(xb), yb)
        loss.backward()
        opt.step()
        r_losses.append(float(loss.detach()))
    o_losses = U.config0_steps(ou, steps=3, lr=0.01)
    assert r_losses == o_losses, (r_losses, o_losses)
    assert r_losses[-1] < r_losses[0]
    torch.manual_seed(1234)   # weights are not stored: the seeded init of the oracle module reproduces the reference's
    ru0, _ = UNet(3, 4), torch.manual_seed(1234)
    ou0 = U.OUNet(3, 4)
    assert all(torch.equal(a, b) for a, b in zip(ru0.state_dict().values(), ou0.state_dict().values()))
    np.savez_compressed(os.path.join(OUT, "unet_config0.npz"), losses=np.array(r_losses), logit_rows=r_logits[:, :, ::16, ::16].numpy(),
                        init_seed=np.int64(1234), n_params=np.int64(sum(p.numel() for p in ru.parameters())))
    print("selectors_ref.npz, unet_config0.npz written; reference executed, oracle == reference")
    print("  ceal orders", conf_order, margin_order, ent_order, "| mc", mc_order, "| core", core_picks, "| unet losses", r_losses)


if __name__ == "__main__":
    main()
