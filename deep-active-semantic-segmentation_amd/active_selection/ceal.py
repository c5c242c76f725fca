"""CEAL softmax-score selection on the HIP path -- mirror of active_selection/ceal.py:13-166.

One deterministic forward per batch; max-probability / top-2 margin / softmax entropy (log2, 1e-12),
the label mask conventions (masked pixels count as 1, 1 and 0 respectively), the per-image mean and
the weak-label argmax all run in the dass_softmax_scores / dass_weak_labels kernels straight from the
NCHW logits -- no numpy argsort on the host (ceal.py:85-91).
"""
import random

import torch

from active_selection.base import ActiveSelectionBase
from dass_hip import ops

_CONF, _MARGIN, _ENTROPY = 0, 1, 2


class ActiveSelectionCEAL(ActiveSelectionBase):

    def __init__(self, dataset_num_classes, dataset_lmdb_env, crop_size, dataloader_batch_size, **kw):
        super(ActiveSelectionCEAL, self).__init__(dataset_lmdb_env, crop_size, dataloader_batch_size, **kw)
        self.dataset_num_classes = dataset_num_classes

    def _scores(self, model, images, mode):
        model.eval()
        local, _ = self.local_slice(images)
        dev = next(self.unwrap(model).parameters()).device
        out = []
        with torch.no_grad():
            for sample in self.make_loader(local, True):
                image_batch = sample['image'].to(dev)
                label_batch = sample['label'].to(dev)
                logits = model(image_batch)
                logits = logits[0] if isinstance(logits, tuple) else logits
                _, means = ops.softmax_scores(logits, label_batch, self.dataset_num_classes, mode)
                out.append(means)
        local_scores = torch.cat(out) if out else torch.zeros((0,), dtype=torch.float32, device=dev)
        return self.gather(local_scores, len(images)).cpu().tolist()

    def get_least_confident_samples(self, model, images, selection_count):
        max_confidence = self._scores(model, images, _CONF)
        return list(zip(*sorted(zip(max_confidence, images), key=lambda x: x[0], reverse=False)))[1][:selection_count]

    def get_least_margin_samples(self, model, images, selection_count):
        margins = self._scores(model, images, _MARGIN)
        return list(zip(*sorted(zip(margins, images), key=lambda x: x[0], reverse=False)))[1][:selection_count]

    def _get_entropies(self, model, images):
        return self._scores(model, images, _ENTROPY)

    def get_maximum_entropy_samples(self, model, images, selection_count):
        entropies = self._get_entropies(model, images)
        selected_samples = list(zip(*sorted(zip(entropies, images), key=lambda x: x[0], reverse=True)))[1][:selection_count]
        return selected_samples, entropies

    def get_fusion_of_confidence_margin_entropy_samples(self, model, images, selection_count):
        samples1 = self.get_least_confident_samples(model, images, selection_count)
        samples2 = self.get_least_margin_samples(model, images, selection_count)
        samples3 = self.get_maximum_entropy_samples(model, images, selection_count)[0]
        samples = list(set(samples1 + samples2 + samples3))
        random.shuffle(samples)
        return samples[:selection_count]

    def get_weakly_labeled_data(self, model, images, threshold, entropies=None):
        if not entropies:
            entropies = self._get_entropies(model, images)
        selected_images = [image for image, entropy in zip(images, entropies) if entropy < threshold]
        weak_labels = []
        dev = next(self.unwrap(model).parameters()).device
        with torch.no_grad():
            for sample in self.make_loader(selected_images, True):
                image_batch = sample['image'].to(dev)
                label_batch = sample['label'].to(dev)
                logits = model(image_batch)
                logits = logits[0] if isinstance(logits, tuple) else logits
                wl = ops.weak_labels(logits, label_batch, self.dataset_num_classes).cpu().numpy()
                weak_labels.extend(wl[i] for i in range(wl.shape[0]))
        return dict(zip(selected_images, weak_labels))
