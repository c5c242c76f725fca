"""MC-dropout vote-entropy selection on the HIP path -- mirror of active_selection/mc_dropout.py:17-196.

Public surface kept: ActiveSelectionMCDropout(num_classes, env, crop_size, batch_size) with
get_random_uncertainity, _get_vote_entropy_for_batch, square_nms, suppress_labeled_entropy,
create_region_maps, get_vote_entropy_for_images.  Selectors still flip Dropout2d modules to train mode
and call model.eval() on exit like the reference.

What runs where: T is `constants.MC_STEPS` read at call time (or the `steps` argument).  For a DeepLab
from this build the T passes share one deterministic prefix (DeepLab.mc_dropout_votes); any other
callable model gets T full forwards + the HIP argmax kernel.  Votes stay uint8 on the device; the
per-class count / p*log2(p+1e-12) / label mask / per-image mean is one kernel; region maps use the
box-sum, min-max and single-workgroup NMS kernels.
"""
import math
import random

import torch

import constants
from active_selection.base import ActiveSelectionBase, merged_batches, score_merge
from dass_hip import ops


def _turn_on_dropout(m):
    if type(m) == torch.nn.Dropout2d:
        m.train()


class ActiveSelectionMCDropout(ActiveSelectionBase):

    def __init__(self, dataset_num_classes, dataset_lmdb_env, crop_size, dataloader_batch_size, **kw):
        super(ActiveSelectionMCDropout, self).__init__(dataset_lmdb_env, crop_size, dataloader_batch_size, **kw)
        self.dataset_num_classes = dataset_num_classes

    def get_random_uncertainity(self, images, selection_count):
        scores = [random.random() for _ in range(len(images))]
        return list(zip(*sorted(zip(scores, images), key=lambda x: x[0], reverse=True)))[1][:selection_count]

    # ------------------------------------------------------------------ votes
    def _votes(self, model, image_batch, steps, masks=None):
        core = self.unwrap(model)
        fast = hasattr(core, "mc_dropout_votes") and not getattr(core.backbone, "mc_dropout", False) \
            and not getattr(core, "noisy_features", False) and core._bn_all_eval()
        if fast:
            return core.mc_dropout_votes(image_batch, steps, masks=masks)
        n, _, h, w = image_batch.shape
        votes = torch.empty((n, steps, h, w), dtype=torch.uint8, device=image_batch.device)
        with torch.no_grad():
            for step in range(steps):
                out = model(image_batch)
                ops.argmax_nchw(out[0] if isinstance(out, tuple) else out, votes, step)
        return votes

    def _get_vote_entropy_for_batch(self, model, image_batch, label_batch, steps=None, masks=None, with_means=False):
        steps = constants.MC_STEPS if steps is None else steps
        votes = self._votes(model, image_batch, steps, masks)
        emap, means = ops.vote_entropy(votes, label_batch, self.dataset_num_classes, want_map=True)
        maps = [emap[i] for i in range(emap.shape[0])]
        return (maps, means) if with_means else maps

    def _image_scores(self, model, images, steps=None):
        """per-image mean vote entropy of THIS rank's shard -> gathered [len(images)] f32 on the device"""
        steps = constants.MC_STEPS if steps is None else steps
        local, _ = self.local_slice(images)
        scores = []
        dev = next(self.unwrap(model).parameters()).device
        core = self.unwrap(model)
        pre = None
        if (type(self)._votes is ActiveSelectionMCDropout._votes  # (a subclass that overrides _votes -- fixed masks in the tests -- keeps its hook)
                and hasattr(core, "mc_prefix") and not getattr(core.backbone, "mc_dropout", False) and not getattr(core, "noisy_features", False)
                and core._bn_all_eval()):
            pre = ops.mc_prefix_stream(dev)
        if pre is None:
            for sample in self.make_loader(local, True):
                image_batch = sample['image'].to(dev, non_blocking=True)
                label_batch = sample['label'].to(dev, non_blocking=True)
                votes = self._votes(model, image_batch, steps)
                _, means = ops.vote_entropy(votes, label_batch, self.dataset_num_classes, want_map=False)
                scores.append(means)
        else:
            # Two-deep software pipeline over the batches: the deterministic prefix (backbone + ASPP: ~100 short launches that
            # leave most CUs idle) of batch i + 1 runs on a second HIP stream under the T stochastic passes of batch i.
            main = torch.cuda.current_stream(dev)

            def start(sample):
                image_batch = sample['image'].to(dev, non_blocking=True)
                label_batch = sample['label'].to(dev, non_blocking=True)
                ready = torch.cuda.Event()
                ready.record(main)                      # the batch is on the device
                pre.wait_event(ready)
                with torch.cuda.stream(pre):
                    state = core.mc_prefix(image_batch)
                    done = torch.cuda.Event()
                    done.record(pre)
                for t in (state[0],) + tuple(x for x in (state[1] or ()) if torch.is_tensor(x)):
                    t.record_stream(main)               # allocated on `pre`, read by the passes on `main`
                image_batch.record_stream(pre)
                return state, done, label_batch

            loader = iter(merged_batches(self.make_loader(local, True), score_merge(self.dataloader_batch_size)))  # (two loader batches per scoring forward)
            nxt = next(loader, None)
            cur = start(nxt) if nxt is not None else None
            while cur is not None:
                nxt = next(loader, None)
                ahead = start(nxt) if nxt is not None else None
                state, done, label_batch = cur
                main.wait_event(done)
                votes = core.mc_tail(state, steps)
                _, means = ops.vote_entropy(votes, label_batch, self.dataset_num_classes, want_map=False)
                scores.append(means)
                cur = ahead
        local_scores = torch.cat(scores) if scores else torch.zeros((0,), dtype=torch.float32, device=dev)
        return self.gather(local_scores, len(images))

    def get_vote_entropy_for_images(self, model, images, selection_count, steps=None):
        model.apply(_turn_on_dropout)
        scores = self._image_scores(model, images, steps).cpu().tolist()   # ONE device->host copy for the pool
        model.eval()
        selected_samples = list(zip(*sorted(zip(scores, images), key=lambda x: x[0], reverse=True)))[1][:selection_count]
        return selected_samples

    # ------------------------------------------------------------------ regions
    @staticmethod
    def square_nms(score_maps, region_size, max_selection_count):
        if not score_maps.is_cuda:
            raise RuntimeError("square_nms runs on the GPU (pass the device tensor; the reference's .cpu() hop is gone)")
        return ops.square_nms(score_maps, region_size, max_selection_count)

    @staticmethod
    def suppress_labeled_entropy(entropy_map, labeled_region):
        if labeled_region:
            maps = entropy_map.unsqueeze(0) if entropy_map.dim() == 2 else entropy_map
            for lr in labeled_region:
                ops.zero_rect(maps, 0, lr[0], lr[0] + lr[2], lr[1], lr[1] + lr[3])

    def create_region_maps(self, model, images, existing_regions, region_size, selection_size, steps=None):
        model.apply(_turn_on_dropout)
        base_size = 512 if self.crop_size == -1 else self.crop_size
        dev = next(self.unwrap(model).parameters()).device
        out_hw = base_size - region_size + 1
        # this rank's shard of the pool (the T-pass forwards are the cost); existing_regions is indexed like `images`
        local, start = self.local_slice(images)
        score_maps = torch.empty((len(local), out_hw, out_hw), dtype=torch.float32, device=dev)
        map_ctr = 0
        for sample in self.make_loader(local, True):
            image_batch = sample['image'].to(dev)
            label_batch = sample['label'].to(dev)
            maps = self._get_vote_entropy_for_batch(model, image_batch, label_batch, steps)
            emaps = torch.stack(maps)
            for i in range(emaps.shape[0]):
                regs = existing_regions[start + map_ctr + i]
                if regs:
                    for lr in regs:
                        ops.zero_rect(emaps, i, lr[0], lr[0] + lr[2], lr[1], lr[1] + lr[3])
            score_maps[map_ctr:map_ctr + emaps.shape[0]] = ops.box_sum(emaps, region_size)
            map_ctr += emaps.shape[0]
        # one min and one max over the WHOLE pool (mc_dropout.py:152-155), then every rank holds every normalised map
        ops.minmax_normalize_(score_maps, self.global_minmax(ops.minmax(score_maps)))
        score_maps = self.gather(score_maps, len(images))
        num_requested_indices = (selection_size * base_size * base_size) / (region_size * region_size)
        regions, num_selected_indices = ops.square_nms(score_maps, region_size, num_requested_indices)
        new_regions = {}
        for i in range(len(regions)):
            if regions[i] != []:
                new_regions[images[i]] = regions[i]
        model.eval()
        return new_regions, num_selected_indices
