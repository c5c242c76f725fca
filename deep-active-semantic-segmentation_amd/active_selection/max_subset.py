"""Representativeness re-ranking ("max subset") on the HIP path -- mirror of active_selection/max_subset.py:12-140.

`get_representative_images` (the branch `active_train.py:452-453` takes for *_image datasets): 2736-d pooled
decoder features of every image and of the candidates (same feature kernel as core-set), then the greedy
facility-location loop (max_subset.py:17-39) entirely on the device -- f64 distance matrix once, then per pick
one column-score kernel, one first-max argmax and one running-min update, the picked index handed over in
device memory.

`get_representative_regions`: the reference pools a region crop with a kernel of the FULL feature-map size
(max_subset.py:60-63,100-101), which current PyTorch rejects ("Output size is too small"), so its result cannot
be pinned; this build raises NotImplementedError for it instead of guessing the intended semantics.
"""
import numpy as np
import torch

from active_selection.base import ActiveSelectionBase
from dass_hip import ops


class ActiveSelectionMaxSubset(ActiveSelectionBase):

    def __init__(self, dataset_lmdb_env, crop_size, dataloader_batch_size, **kw):
        super(ActiveSelectionMaxSubset, self).__init__(dataset_lmdb_env, crop_size, dataloader_batch_size, **kw)

    def _max_representative_samples(self, image_features, candidate_image_features, selection_count):
        def dev(f):
            t = f if torch.is_tensor(f) else torch.as_tensor(np.asarray(f, dtype=np.float32))
            return t if t.is_cuda else t.cuda()

        print('Finding max representative candidates..')
        return ops.max_representative(dev(image_features), dev(candidate_image_features), selection_count).cpu().tolist()

    def _convert_regions_to_list(self, regions):
        list_images, list_regions = [], []
        for ir in sorted(list(regions.keys())):
            for r in regions[ir]:
                list_images.append(ir)
                list_regions.append(r)
        return list_images, list_regions

    def _get_features_for_images(self, model, images):
        core = self.unwrap(model)
        local, _ = self.local_slice(images)
        dev = next(core.parameters()).device
        rows = []
        model.eval()
        core.set_return_features(True)
        with torch.no_grad():
            for sample in self.make_loader(local, False):
                batch = sample['image'] if isinstance(sample, dict) else sample
                rows.append(ops.avgpool_features(core.encoder_features(batch.to(dev)), 64, 32))
        core.set_return_features(False)
        feats = torch.cat(rows) if rows else torch.zeros((0, 2736), dtype=torch.float32, device=dev)
        return self.gather(feats, len(images))

    def get_representative_images(self, model, all_images, candidate_images):
        print('Getting features for images for representativeness ..')
        all_image_features = self._get_features_for_images(model, all_images)
        candidate_features = self._get_features_for_images(model, candidate_images)
        selected_candidate_indices = self._max_representative_samples(all_image_features, candidate_features,
                                                                      len(candidate_images) // 2)
        return [candidate_images[i] for i in selected_candidate_indices]

    def get_representative_regions(self, model, all_images, candidate_regions, region_size):
        raise NotImplementedError("reference pools region crops with a full-map kernel (max_subset.py:60-63), which "
                                  "errors in current PyTorch; its intended result cannot be pinned")
