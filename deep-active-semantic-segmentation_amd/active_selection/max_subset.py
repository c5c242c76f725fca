"""Representativeness re-ranking ("max subset") on the HIP path -- mirror of active_selection/max_subset.py:12-140.

`get_representative_images` (the branch `active_train.py:452-453` takes for *_image datasets): 2736-d pooled
decoder features of every image and of the candidates (same feature kernel as core-set), then the greedy
facility-location loop (max_subset.py:17-39) entirely on the device -- f64 distance matrix once, then per pick
one column-score kernel, one first-max argmax and one running-min update, the picked index handed over in
device memory.

`get_representative_regions` (the `variance_representative` region branch, active_train.py:445-514): every image's
feature map is cut into a grid of region-sized cells and every candidate region into its feature-map crop, each reduced
to ONE 304-vector, then the same greedy loop ranks the candidate regions.  As written the reference reduces a cell with
`F.avg_pool2d(cell, (H_feat, W_feat))` -- a kernel of the FULL feature-map size over an h x w crop (max_subset.py:62-63,
110-111) -- which every PyTorch rejects ("Output size is too small"), so that code can never have produced a number and
no fixture can be generated from it.  This build implements the evident intent, the average over the cell (for the grid:
avg_pool with kernel = stride = cell size; for a candidate region: the mean of its crop), and pins it against
oracle/selection_cpu.py:region_grid_features / region_features (parity with the reference itself: UNPINNED for this one
function, for the reason above).
"""
import math

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

    def _feature_batches(self, model, images):
        """yields (first index into `images`, [B,304,h,w] decoder features) for THIS rank's shard"""
        core = self.unwrap(model)
        local, start = self.local_slice(images)
        dev = next(core.parameters()).device
        model.eval()
        core.set_return_features(True)
        try:
            with torch.no_grad():
                for sample in self.make_loader(local, False):
                    batch = sample['image'] if isinstance(sample, dict) else sample
                    yield start, core.encoder_features(batch.to(dev))
                    start += batch.shape[0]
        finally:
            core.set_return_features(False)

    def _get_features_for_image_regions(self, model, images, region_size):
        """max_subset.py:49-71: one 304-vector per grid cell, cells in (image, row, col) order -> [len * rows * cols, 304]"""
        rows_out, per_image = [], 1
        dev = next(self.unwrap(model).parameters()).device
        for _, feats in self._feature_batches(model, images):
            b, c, hh, ww = feats.shape
            h = math.floor(region_size * hh / self.crop_size)
            w = math.floor(region_size * ww / self.crop_size)
            nr, nc = math.floor(hh / h), math.floor(ww / w)
            per_image = nr * nc
            if h == w:  # square cells: the pooled-feature kernel with kernel = stride = cell (channel-major rows)
                pooled = ops.avgpool_features(feats, h, h).view(b, c, nr, nc)  # (hh - h) // h + 1 == floor(hh / h) cells
                rows_out.append(pooled.permute(0, 2, 3, 1).reshape(b * nr * nc, c).float())
            else:
                cells = [ops.global_avgpool(feats[:, :, r * h:r * h + h, q * w:q * w + w].contiguous(memory_format=torch.channels_last)).reshape(b, 1, c)
                         for r in range(nr) for q in range(nc)]
                rows_out.append(torch.cat(cells, dim=1).reshape(b * nr * nc, c).float())
        local = torch.cat(rows_out) if rows_out else torch.zeros((0, 304), dtype=torch.float32, device=dev)
        if not self.shard:
            return local
        from active_selection.base import _dist, all_gather_rows

        dist = _dist()
        if dist is not None and dist.get_world_size() > 1:
            # a rank whose shard is empty (fewer images than ranks) never saw a feature map: the cells-per-image count is agreed
            # on by all ranks before the gather, or its padded buffer would have another shape than its peers'
            t = torch.tensor([per_image], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            per_image = int(t.item())
        full = all_gather_rows(local.view(-1, per_image, local.shape[1]), len(images))  # image-granular shards
        return full.reshape(-1, local.shape[1])

    def _get_features_for_regions(self, model, list_images, list_regions):
        """max_subset.py:91-113: region (r, c, h, w) of image i -> mean of its feature-map crop -> [len, 304]"""
        rows_out = []
        dev = next(self.unwrap(model).parameters()).device
        for start, feats in self._feature_batches(model, list_images):
            rr, rc = feats.shape[2] / self.crop_size, feats.shape[3] / self.crop_size
            for i in range(feats.shape[0]):
                r0, c0, h0, w0 = list_regions[start + i]
                r, c, h, w = math.floor(r0 * rr), math.floor(c0 * rc), math.floor(h0 * rr), math.floor(w0 * rc)
                crop = feats[i:i + 1, :, r:r + h, c:c + w].contiguous(memory_format=torch.channels_last)
                rows_out.append(ops.global_avgpool(crop).reshape(1, -1).float())
        local = torch.cat(rows_out) if rows_out else torch.zeros((0, 304), dtype=torch.float32, device=dev)
        return self.gather(local, len(list_images))

    def get_representative_regions(self, model, all_images, candidate_regions, region_size):
        candidate_list_images, candidate_list_regions = self._convert_regions_to_list(candidate_regions)
        print('Getting features for images for representativeness ..')
        all_image_features = self._get_features_for_image_regions(model, all_images, region_size)
        print('Getting features for candidates for representativeness ..')
        region_features = self._get_features_for_regions(model, candidate_list_images, candidate_list_regions)
        selected_candidate_indices = self._max_representative_samples(all_image_features, region_features,
                                                                      len(region_features) // 2)
        selected_regions = {}
        for i in selected_candidate_indices:
            selected_regions.setdefault(candidate_list_images[i], []).append(candidate_list_regions[i])
        return selected_regions, len(selected_candidate_indices)
