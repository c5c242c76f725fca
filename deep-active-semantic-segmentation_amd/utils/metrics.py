"""utils.metrics.Evaluator -- the surface of utils/metrics.py:6-49 with the confusion matrix accumulated ON THE DEVICE.

Same constructor, methods (Pixel_Accuracy, Pixel_Accuracy_Class, Mean_Intersection_over_Union,
Frequency_Weighted_Intersection_over_Union, add_batch, reset) and `.confusion_matrix` as a float64 numpy array with
rows = ground truth, columns = prediction.  `add_batch(gt_image, pre_image)` takes what `Trainer.validation` hands it
(active_train.py:159-163: numpy label map + numpy argmax map) as well as device tensors; `pre_image` may also be the raw
NCHW logits, in which case argmax + histogram are ONE kernel and only the final C x C matrix ever leaves the GPU.
There is no host-side histogram: inputs are moved to the GPU and counted by dass_confusion_accumulate.
"""
import numpy as np
import torch


class Evaluator(object):

    def __init__(self, num_class):
        np.seterr(divide='ignore', invalid='ignore')
        self.num_class = num_class
        self._dev = None
        self._base = np.zeros((num_class, num_class))  # a matrix assigned through the setter

    # ---- confusion matrix
    @property
    def confusion_matrix(self):
        if self._dev is None:
            return self._base
        return self._base + self._dev.cpu().numpy().astype(np.float64)

    @confusion_matrix.setter
    def confusion_matrix(self, value):
        self._base = np.asarray(value, dtype=np.float64)
        self._dev = None

    def reset(self):
        self.confusion_matrix = np.zeros((self.num_class, self.num_class))

    def add_batch(self, gt_image, pre_image):
        from dass_hip import ops

        if not torch.cuda.is_available():
            raise RuntimeError("Evaluator.add_batch counts on the GPU (dass_confusion_accumulate); there is no CPU fallback")
        gt = gt_image if torch.is_tensor(gt_image) else torch.from_numpy(np.ascontiguousarray(gt_image))
        pre = pre_image if torch.is_tensor(pre_image) else torch.from_numpy(np.ascontiguousarray(pre_image))
        dev = pre.device if pre.is_cuda else (gt.device if gt.is_cuda else torch.device("cuda", torch.cuda.current_device()))
        gt, pre = gt.to(dev), pre.to(dev)
        if pre.dim() == gt.dim():
            assert gt.shape == pre.shape
        if self._dev is None or self._dev.device != dev:
            if self._dev is not None:
                self._base = self.confusion_matrix
            self._dev = torch.zeros((self.num_class, self.num_class), dtype=torch.int64, device=dev)
        ops.confusion_accumulate(self._dev, gt, pre, self.num_class)

    # ---- metrics over the matrix (host arithmetic on num_class^2 numbers)
    def _parts(self):
        cm = self.confusion_matrix
        hit = np.diag(cm)
        gt_count, pred_count = cm.sum(axis=1), cm.sum(axis=0)
        return cm, hit, gt_count, pred_count

    def Pixel_Accuracy(self):
        cm, hit, _, _ = self._parts()
        return hit.sum() / cm.sum()

    def Pixel_Accuracy_Class(self):
        _, hit, gt_count, _ = self._parts()
        return np.nanmean(hit / gt_count)

    def _iou(self):
        _, hit, gt_count, pred_count = self._parts()
        return hit / (gt_count + pred_count - hit)

    def Mean_Intersection_over_Union(self):
        return np.nanmean(self._iou())

    def Frequency_Weighted_Intersection_over_Union(self):
        cm, _, gt_count, _ = self._parts()
        share = gt_count / cm.sum()
        seen = share > 0
        return (share[seen] * self._iou()[seen]).sum()
