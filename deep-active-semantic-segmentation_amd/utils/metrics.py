"""utils.metrics.Evaluator -- mirror of utils/metrics.py:6-49 with the confusion matrix kept on the device.

Same constructor, methods and formulas (Pixel_Accuracy, Pixel_Accuracy_Class, Mean_Intersection_over_Union,
Frequency_Weighted_Intersection_over_Union, add_batch, reset, .confusion_matrix as a float64 numpy array).
`add_batch(gt_image, pre_image)` keeps the reference's numpy path for numpy inputs (host bookkeeping,
active_train.py:159-163) and accepts device tensors: `pre_image` may be the argmax map OR the raw NCHW logits, in
which case argmax + histogram run in one kernel and nothing but the final C x C matrix ever leaves the GPU.
"""
import numpy as np
import torch


class Evaluator(object):

    def __init__(self, num_class):
        np.seterr(divide='ignore', invalid='ignore')
        self.num_class = num_class
        self._host = np.zeros((self.num_class,) * 2)
        self._dev = None

    @property
    def confusion_matrix(self):
        if self._dev is not None:
            return self._host + self._dev.cpu().numpy().astype(np.float64)
        return self._host

    @confusion_matrix.setter
    def confusion_matrix(self, value):
        self._host = np.asarray(value, dtype=np.float64)
        self._dev = None

    def Pixel_Accuracy(self):
        cm = self.confusion_matrix
        return np.diag(cm).sum() / cm.sum()

    def Pixel_Accuracy_Class(self):
        cm = self.confusion_matrix
        return np.nanmean(np.divide(np.diag(cm), cm.sum(axis=1)))

    def Mean_Intersection_over_Union(self):
        cm = self.confusion_matrix
        return np.nanmean(np.divide(np.diag(cm), (np.sum(cm, axis=1) + np.sum(cm, axis=0) - np.diag(cm))))

    def Frequency_Weighted_Intersection_over_Union(self):
        cm = self.confusion_matrix
        freq = np.sum(cm, axis=1) / np.sum(cm)
        iu = np.divide(np.diag(cm), (np.sum(cm, axis=1) + np.sum(cm, axis=0) - np.diag(cm)))
        return (freq[freq > 0] * iu[freq > 0]).sum()

    def _generate_matrix(self, gt_image, pre_image):
        mask = (gt_image >= 0) & (gt_image < self.num_class)
        label = self.num_class * gt_image[mask].astype('int') + pre_image[mask]
        count = np.bincount(label, minlength=self.num_class**2)
        return count.reshape(self.num_class, self.num_class)

    def add_batch(self, gt_image, pre_image):
        if torch.is_tensor(pre_image) and pre_image.is_cuda:
            from dass_hip import ops

            if self._dev is None:
                self._dev = torch.zeros((self.num_class, self.num_class), dtype=torch.int64, device=pre_image.device)
            ops.confusion_accumulate(self._dev, gt_image, pre_image, self.num_class)
            return
        assert gt_image.shape == pre_image.shape
        self._host = self._host + self._generate_matrix(gt_image, pre_image)

    def reset(self):
        self._host = np.zeros((self.num_class,) * 2)
        self._dev = None
