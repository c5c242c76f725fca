"""utils.loss.SegmentationLosses on the HIP path -- mirror of utils/loss.py:5-70.

Same constructor, build_loss(mode) -> bound callable (logit[N,C,H,W], target[N,H,W] float) -> 0-dim
tensor with autograd, NotImplementedError for unknown modes.  The log-softmax + NLL + ignore-index mean
(and its gradient) are the fused dass_ce_* kernels; the scalar focal transform stays scalar torch math,
exactly as the reference composes it on top of the CE scalar.

Multi-process training (one process per GPU).  The reference evaluates the loss on the logits GATHERED from all DataParallel
replicas and divides by the GLOBAL batch size (utils/loss.py:39-51 behind active_train.py:82-85,104-105).  That form is an
explicit opt-in here -- SegmentationLosses(..., global_batch=True) -- because it issues collectives inside every loss call:
  * contract: EVERY rank calls the loss the same number of times, and the gradients are then AVERAGED over the ranks
    (dass_hip.dist.GradientAverager / average_gradients, torch DDP): the backward pre-multiplies by the world size so that
    the averaged gradient equals the single-process one;
  * calls with autograd disabled (validation under torch.no_grad(), where ranks may hold different numbers of batches) stay
    rank-local unless global_batch_in_eval=True;
  * the default (global_batch=False) never communicates: a process group that exists only for sharded pool scoring does not
    change what a rank-local training step computes;
  * `use_static_global(device)` moves the exchange OUT of the loss call: the denominators depend on the labels only, so
    `criterion.static.exchange(target)` runs eagerly before the step and the call itself holds no collective -- what a step replayed
    from a hipGraph needs (dass_hip/graph.py; 'ce' only: the focal transform is not linear in a rank's share of the loss).
"""
import torch

from dass_hip import ops
from dass_hip.dist import collectives_on, global_batch_mean, sum_over_ranks


class SegmentationLosses(object):

    def __init__(self, weight=None, batch_average=True, ignore_index=255, cuda=True, global_batch=False, global_batch_in_eval=False):
        self.ignore_index = ignore_index
        self.weight = weight
        self.batch_average = batch_average
        self.cuda = cuda
        self.global_batch = global_batch
        self.global_batch_in_eval = global_batch_in_eval
        self.static = None

    def use_static_global(self, device):
        """-> dass_hip.dist.StaticGlobalBatch whose `exchange(target)` the caller runs before every (graphed) step"""
        from dass_hip.dist import StaticGlobalBatch

        self.static = StaticGlobalBatch(device, self.ignore_index, self._weight_on(device))
        return self.static

    def _global(self):
        """does this call exchange its numerator / counts with the other ranks?"""
        return bool(self.global_batch) and collectives_on() and (torch.is_grad_enabled() or self.global_batch_in_eval)

    def build_loss(self, mode='ce'):
        if mode == 'ce':
            return self.CrossEntropyLoss
        elif mode == 'focal':
            return self.FocalLoss
        else:
            raise NotImplementedError

    def _weight_on(self, device):
        if self.weight is None:
            return None
        w = self.weight if torch.is_tensor(self.weight) else torch.as_tensor(self.weight)
        return w.to(device=device, dtype=torch.float32)

    def _ce_mean(self, logit, target):
        """-> (CE mean over the valid pixels of the GLOBAL batch, global batch size).  The reference evaluates the loss
        on the gathered logits of all DataParallel replicas (active_train.py:104-105), so under one-process-per-GPU the
        ranks exchange numerator, valid-pixel count and image count (dass_hip/dist.py:global_batch_mean); a single
        process takes the fused mean kernel path unchanged."""
        n = logit.size(0)
        if not self._global():
            return ops.cross_entropy(logit, target, self._weight_on(logit.device), self.ignore_index), n
        s, cnt = ops.cross_entropy_parts(logit, target, self._weight_on(logit.device), self.ignore_index)
        if self.static is not None and torch.is_grad_enabled():
            return s * self.static.inv_count, self.static.n_global   # no collective in the call (see use_static_global)
        return global_batch_mean(s, cnt, n)

    def SampleWeightedCrossEntropyLoss(self, logit, target, sample_weights):
        n, c, h, w = logit.size()
        weights = sample_weights.to(logit.device)
        # reduction='none' then .mean(-1).mean(-1): ignored pixels count as zeros in the H*W mean
        per_image = torch.stack([ops.cross_entropy_sum(logit[i:i + 1], target[i:i + 1], self._weight_on(logit.device),
                                                       self.ignore_index) for i in range(n)]) / float(h * w)
        if self._global():  # mean over the GLOBAL batch (the reference sees the gathered batch on device 0)
            ng = int(round(float(sum_over_ranks(torch.tensor(float(n), device=logit.device), differentiable=False))))
            loss = sum_over_ranks(torch.sum(torch.mul(per_image, weights))) / ng
            n = ng
        else:
            loss = torch.mean(torch.mul(per_image, weights))
        if self.batch_average:
            loss /= n
        return loss

    def CrossEntropyLoss(self, logit, target):
        loss, n = self._ce_mean(logit, target)
        if self.batch_average:
            loss = loss / n
        return loss

    def FocalLoss(self, logit, target, gamma=2, alpha=0.5):
        if self.static is not None and self._global():
            raise NotImplementedError("focal loss with pre-exchanged denominators: the transform is not linear in a rank's share")
        ce, n = self._ce_mean(logit, target)
        logpt = -ce
        pt = torch.exp(logpt)
        if alpha is not None:
            logpt = logpt * alpha
        loss = -((1 - pt) ** gamma) * logpt
        if self.batch_average:
            loss = loss / n
        return loss
