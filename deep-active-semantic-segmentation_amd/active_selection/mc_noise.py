"""Noise-based vote-entropy selection on the HIP path -- mirror of active_selection/mc_noise.py:16-186.

Same reduction as MC-dropout (votes of T argmax maps -> per-class fractions -> -sum p log2(p+1e-12) -> label mask
-> per-image sum / (H*W)), with the stochasticity coming from gaussian input noise (sigma 0.125, mc_noise.py:24),
from the model's own feature noise (DeepLab.set_noisy_features, deeplab.py:39-56) or from both noise and dropout.
T = constants.MC_STEPS read at call time.  Noise draws come from torch's device RNG (the reference uses numpy on
the host), so these selectors are distribution-equivalent by default; with `noise_source=draw(shape, sigma)` (and the
same hook on the model, DeepLab.noise_source) the draws are supplied by the caller -- the tests replay numpy's seeded
stream and compare votes and entropy maps with the reference's (tests/golden/mc_noise.npz).  Every pass is a full forward.
"""
import torch

import constants
from active_selection.base import ActiveSelectionBase
from active_selection.mc_dropout import ActiveSelectionMCDropout, _turn_on_dropout
from dass_hip import ops


class ActiveSelectionMCNoise(ActiveSelectionBase):

    def __init__(self, num_classes, dataset_lmdb_env, crop_size, dataloader_batch_size, noise_source=None, **kw):
        super(ActiveSelectionMCNoise, self).__init__(dataset_lmdb_env, crop_size, dataloader_batch_size, **kw)
        self.dataset_num_classes = num_classes
        self.noise_source = noise_source

    def _vote_maps(self, model, image_batch, label_batch, perturb=None):
        steps = constants.MC_STEPS
        n, _, h, w = image_batch.shape
        votes = torch.empty((n, steps, h, w), dtype=torch.uint8, device=image_batch.device)
        with torch.no_grad():
            for step in range(steps):
                out = model(perturb(image_batch) if perturb is not None else image_batch)
                ops.argmax_nchw(out[0] if isinstance(out, tuple) else out, votes, step)
        emap, _ = ops.vote_entropy(votes, label_batch, self.dataset_num_classes, want_map=True)
        return [emap[i] for i in range(n)]

    def _get_vote_entropy_for_batch_with_input_noise(self, model, image_batch, label_batch):
        def perturb(x):  # mc_noise.py:24-25: N(0, 0.125) on the normalised image
            if self.noise_source is not None:
                return x + self.noise_source(tuple(x.shape), 0.125).to(device=x.device, dtype=x.dtype)
            return x + torch.randn_like(x) * 0.125

        return self._vote_maps(model, image_batch, label_batch, perturb=perturb)

    def _get_vote_entropy_for_batch_with_feature_noise(self, model, image_batch, label_batch):
        core = self.unwrap(model)
        core.set_noisy_features(True)
        saved = getattr(core, "noise_source", None)
        if self.noise_source is not None:
            core.noise_source = self.noise_source
        try:
            return self._vote_maps(model, image_batch, label_batch)
        finally:
            core.set_noisy_features(False)
            core.noise_source = saved

    def _get_vote_entropy_for_batch_with_mc_dropout(self, model, image_batch, label_batch):
        model.apply(_turn_on_dropout)
        try:
            return self._vote_maps(model, image_batch, label_batch)
        finally:
            model.eval()

    def _scores(self, model, images, per_batch):
        local, _ = self.local_slice(images)
        dev = next(self.unwrap(model).parameters()).device
        model.eval()
        out = []
        for sample in self.make_loader(local, True):
            image_batch, label_batch = sample['image'].to(dev), sample['label'].to(dev)
            maps = per_batch(image_batch, label_batch)
            out.append(torch.stack([m.sum() for m in maps]) / float(image_batch.shape[2] * image_batch.shape[3]))
        local_scores = torch.cat(out) if out else torch.zeros((0,), dtype=torch.float32, device=dev)
        return self.gather(local_scores, len(images)).cpu().tolist()

    @staticmethod
    def _top(entropies, images, selection_count):
        return list(zip(*sorted(zip(entropies, images), key=lambda x: x[0], reverse=True)))[1][:selection_count]

    def get_vote_entropy_for_images_with_input_noise(self, model, images, selection_count):
        ent = self._scores(model, images, lambda x, y: self._get_vote_entropy_for_batch_with_input_noise(model, x, y))
        return self._top(ent, images, selection_count)

    def get_vote_entropy_for_images_with_feature_noise(self, model, images, selection_count):
        ent = self._scores(model, images, lambda x, y: self._get_vote_entropy_for_batch_with_feature_noise(model, x, y))
        return self._top(ent, images, selection_count)

    def _combined(self, model, image_batch, label_batch):
        noise = self._get_vote_entropy_for_batch_with_feature_noise(model, image_batch, label_batch)
        mc = self._get_vote_entropy_for_batch_with_mc_dropout(model, image_batch, label_batch)
        return [a + b for a, b in zip(noise, mc)]

    def get_vote_entropy_for_batch_with_noise_and_vote_entropy(self, model, images, selection_count):
        ent = self._scores(model, images, lambda x, y: self._combined(model, x, y))
        return self._top(ent, images, selection_count)

    def create_region_maps(self, model, images, existing_regions, region_size, selection_size):
        base_size = 512 if self.crop_size == -1 else self.crop_size
        dev = next(self.unwrap(model).parameters()).device
        out_hw = base_size - region_size + 1
        local, start = self.local_slice(images)  # sharded like ActiveSelectionMCDropout.create_region_maps
        score_maps = torch.empty((len(local), out_hw, out_hw), dtype=torch.float32, device=dev)
        map_ctr = 0
        for sample in self.make_loader(local, True):
            image_batch, label_batch = sample['image'].to(dev), sample['label'].to(dev)
            emaps = torch.stack(self._combined(model, image_batch, label_batch))
            for i in range(emaps.shape[0]):
                for lr in existing_regions[start + map_ctr + i] or []:
                    ops.zero_rect(emaps, i, lr[0], lr[0] + lr[2], lr[1], lr[1] + lr[3])
            score_maps[map_ctr:map_ctr + emaps.shape[0]] = ops.box_sum(emaps, region_size)
            map_ctr += emaps.shape[0]
        ops.minmax_normalize_(score_maps, self.global_minmax(ops.minmax(score_maps)))
        score_maps = self.gather(score_maps, len(images))
        num_requested_indices = (selection_size * base_size * base_size) / (region_size * region_size)
        regions, num_selected_indices = ActiveSelectionMCDropout.square_nms(score_maps, region_size, num_requested_indices)
        new_regions = {images[i]: regions[i] for i in range(len(regions)) if regions[i] != []}
        model.eval()
        return new_regions, num_selected_indices
