import os
"""Core-set (k-center greedy) selection on the HIP path -- mirror of active_selection/core_set.py:12-69.

Feature extraction stops at the 304-channel decoder feature map (the reference also runs last_conv and
the x4 upsample and throws them away, core_set.py:60), pools it with the avg_pool(64, 32) kernel into
the channel-major 2736-vector, all-gathers the shards and runs the greedy loop on the device:
f64 distances to the newest centre, running min, first-max argmax -- every pick is three small
launches with the picked index handed over in device memory, one host copy at the end.
"""
import numpy as np
import torch

from active_selection.base import ActiveSelectionBase, merged_batches, score_merge
from dass_hip import ops


class ActiveSelectionCoreSet(ActiveSelectionBase):

    def __init__(self, dataset_lmdb_env, crop_size, dataloader_batch_size, **kw):
        super(ActiveSelectionCoreSet, self).__init__(dataset_lmdb_env, crop_size, dataloader_batch_size, **kw)

    def _select_batch(self, features, selected_indices, N):
        feats = features if torch.is_tensor(features) else torch.as_tensor(np.asarray(features, dtype=np.float32))
        if not feats.is_cuda:
            feats = feats.cuda()
        picks, min_dist = ops.kcenter_greedy(feats, list(selected_indices), N)
        new_batch = picks.cpu().tolist()
        assert not set(new_batch) & set(selected_indices)
        print('Maximum distance from cluster centers is %0.5f' % float(min_dist.max()))
        return new_batch

    def _updated_distances(self, cluster_centers, features, min_distances):
        """core_set.py:32-38: distance of every row to its nearest centre among `cluster_centers`, folded into the running
        minimum (None = first call) -- one dass_kcenter_update launch per centre.  -> float64 numpy [N, 1] like the
        reference.  (_select_batch keeps the whole greedy loop on the device and does not go through the host here.)"""
        feats = features if torch.is_tensor(features) else torch.as_tensor(np.asarray(features, dtype=np.float32))
        feats = (feats if feats.is_cuda else feats.cuda()).contiguous().float()
        n = feats.shape[0]
        if min_distances is None:
            md = torch.full((n,), float("inf"), dtype=torch.float64, device=feats.device)
        else:
            md = torch.as_tensor(np.asarray(min_distances, dtype=np.float64).reshape(-1)).to(feats.device)
        ops.kcenter_update(feats, list(cluster_centers), md)
        return md.cpu().numpy().reshape(-1, 1)

    def _features(self, model, paths):
        core = self.unwrap(model)
        if core.model_name == 'deeplab':
            feature_dim, k = 2736, 64
        else:
            raise NotImplementedError("only the DeepLab feature tap is on this build's path")
        local, _ = self.local_slice(paths)
        dev = next(core.parameters()).device
        rows = []
        model.eval()
        # The batches are independent and one encoder pass is ~100 short launches that leave most CUs idle: the batches are dealt
        # over DASS_CORESET_LANES (default 2; 3-4 measured 1040-1320 depending on how the streams map to hardware queues) HIP streams (DASS_MC_PIPELINE=0: one stream)
        lanes = [None] + ops.extra_streams(dev, int(os.environ.get("DASS_CORESET_LANES", "2")) - 1)
        main = torch.cuda.current_stream(dev)
        used_side = False
        with torch.no_grad():
            for i, sample in enumerate(merged_batches(self.make_loader(local, False), score_merge(self.dataloader_batch_size, most=3))):
                batch = (sample['image'] if isinstance(sample, dict) else sample).to(dev)
                side = lanes[i % len(lanes)] if i > 0 else None
                if side is not None:
                    ready = torch.cuda.Event()
                    ready.record(main)   # the batch is on the device; operand caches filled by batch 0 are complete
                    side.wait_event(ready)
                    with torch.cuda.stream(side):
                        pooled = ops.avgpool_features(core.encoder_features(batch), k, k // 2)
                    pooled.record_stream(main)
                    batch.record_stream(side)
                    used_side = True
                else:
                    pooled = ops.avgpool_features(core.encoder_features(batch), k, k // 2)
                assert pooled.shape[1] == feature_dim, pooled.shape
                rows.append(pooled)
            if used_side:
                for st in lanes[1:]:
                    done = torch.cuda.Event()
                    done.record(st)
                    main.wait_event(done)
        local_feats = torch.cat(rows) if rows else torch.zeros((0, feature_dim), dtype=torch.float32, device=dev)
        return self.gather(local_feats, len(paths))

    def get_k_center_greedy_selections(self, selection_size, model, candidate_image_batch, already_selected_image_batch):
        combined_paths = already_selected_image_batch + candidate_image_batch
        core = self.unwrap(model)
        core.set_return_features(True)   # kept for parity with core_set.py:52,67 (callers may inspect it)
        features = self._features(model, combined_paths)
        core.set_return_features(False)
        selected_indices = self._select_batch(features, list(range(len(already_selected_image_batch))), selection_size)
        return [combined_paths[i] for i in selected_indices]
