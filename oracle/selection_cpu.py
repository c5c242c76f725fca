"""CPU oracle: losses and acquisition scoring in stock PyTorch / numpy.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates, with T and the dropout masks as parameters:
  utils/loss.py:22-70                     CE / focal / sample-weighted CE
  active_selection/mc_dropout.py:30-49    vote entropy of T argmax maps (log2, 1e-12, label mask -> 0)
  active_selection/mc_dropout.py:82-155   labeled-region suppression, box sums, global min-max, square NMS
  active_selection/mc_dropout.py:189-195  per-image mean + stable descending sort + top-k
  active_selection/ceal.py:34-39,82-95,111-123,158-164   confidence / margin / entropy / weak labels
  active_selection/core_set.py:17-38,56-63 feature pooling + k-center greedy (sklearn fp64 distances)
  active_selection/max_subset.py:17-39,49-113  greedy facility location, region features
  utils/metrics.py:6-49                   confusion matrix + the four metrics
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ losses (utils/loss.py)
def ce_loss(logit, target, weight=None, ignore_index=255, batch_average=True):
    loss = F.cross_entropy(logit, target.long(), weight=weight, ignore_index=ignore_index, reduction="mean")
    return loss / logit.shape[0] if batch_average else loss


def focal_loss(logit, target, weight=None, ignore_index=255, batch_average=True, gamma=2, alpha=0.5):
    logpt = -F.cross_entropy(logit, target.long(), weight=weight, ignore_index=ignore_index, reduction="mean")
    pt = torch.exp(logpt)
    if alpha is not None:
        logpt = logpt * alpha
    loss = -((1 - pt) ** gamma) * logpt
    return loss / logit.shape[0] if batch_average else loss


def sample_weighted_ce_loss(logit, target, sample_weights, weight=None, ignore_index=255, batch_average=True):
    per_pix = F.cross_entropy(logit, target.long(), weight=weight, ignore_index=ignore_index, reduction="none")
    loss = torch.mean(per_pix.mean(-1).mean(-1) * sample_weights)
    return loss / logit.shape[0] if batch_average else loss


# ------------------------------------------------------------------ MC-dropout vote entropy
def label_mask(label, num_classes):
    return (label < 0) | (label >= num_classes)


def vote_entropy_maps(votes, label, num_classes):
    """votes [B,T,H,W] (any numeric dtype), label [B,H,W] float -> list of B [H,W] f32 maps"""
    votes = votes.float()
    t = votes.shape[1]
    maps = []
    for i in range(votes.shape[0]):
        e = torch.zeros(votes.shape[2], votes.shape[3])
        for c in range(num_classes):
            p = torch.sum(votes[i] == c, dim=0, dtype=torch.float32) / t
            e = e - (p * torch.log2(p + 1e-12))
        if label is not None:
            e[label_mask(label[i], num_classes)] = 0
        maps.append(e)
    return maps


def mc_votes(model, image, masks_t):
    """T stochastic forwards the reference way (full forward per pass): -> [B,T,H,W] argmax votes"""
    m1, m2 = masks_t
    with torch.no_grad():
        return torch.stack([torch.argmax(model(image, (m1[t], m2[t])), dim=1) for t in range(m1.shape[0])], dim=1)


def select_top(scores, keys, count, reverse=True):
    """list(zip(*sorted(zip(scores, keys), key=score, reverse=...)))[1][:count] (stable)"""
    order = sorted(zip(scores, keys), key=lambda x: x[0], reverse=reverse)
    return [k for _, k in order][:count]


# ------------------------------------------------------------------ CEAL scores
def softmax_score_maps(logits, label, num_classes):
    """-> (confidence, margin, entropy) maps [B,H,W] f32 with the reference's mask conventions"""
    sm = torch.softmax(logits, dim=1)
    mask = label_mask(label, num_classes)
    conf = sm.max(dim=1)[0].clone()
    conf[mask] = 1
    srt = np.sort(sm.numpy(), axis=1)
    margin = torch.from_numpy(srt[:, -1] - srt[:, -2]).clone()
    margin[mask] = 1
    ent = torch.zeros_like(conf)
    for c in range(sm.shape[1]):
        ent = ent - sm[:, c] * torch.log2(sm[:, c] + 1e-12)
    ent[mask] = 0
    return conf, margin, ent


def weak_label_maps(logits, label, num_classes):
    pred = np.argmax(logits.numpy(), axis=1).astype(np.uint8)
    pred[label_mask(label, num_classes).numpy()] = 255
    return pred


# ------------------------------------------------------------------ core-set
def coreset_features(feats, k=64):
    """[B,304,h,w] -> [B,2736] : avg_pool2d(k, k//2) flattened channel-major, as float64 rows"""
    pooled = F.avg_pool2d(feats, (k, k), k // 2)
    return pooled.reshape(pooled.shape[0], -1).numpy().astype(np.float64)


def kcenter_greedy(features, selected, count):
    from sklearn.metrics import pairwise_distances

    def upd(centers, cur):
        d = pairwise_distances(features, features[centers, :], metric="euclidean")
        return np.min(d, axis=1).reshape(-1, 1) if cur is None else np.minimum(cur, d)

    md = upd(list(selected), None)
    picks = []
    for _ in range(count):
        i = int(np.argmax(md))
        md = upd([i], md)
        picks.append(i)
    return picks, float(md.max())


# ------------------------------------------------------------------ regions
def suppress_labeled(entropy_map, regions):
    for (r, c, h, w) in regions or []:
        entropy_map[r:r + h, c:c + w] = 0
    return entropy_map


def box_sum(entropy_map, r):
    return F.conv2d(entropy_map[None, None], torch.ones(1, 1, r, r))[0, 0]


def minmax_normalize(score_maps):
    lo, hi = score_maps.min(), score_maps.max()
    return score_maps.add_(-lo).mul_(1.0 / (hi - lo))


def square_nms(score_maps, region_size, max_selection_count):
    """greedy: global first-argmax, record (row, col, r, r), zero the clipped box
    [row-r, row+r) x [col-r, col+r) of that image, stop once the global max < 0.01"""
    score_maps = score_maps.clone()
    n, h, w = score_maps.shape
    selected = [[] for _ in range(n)]
    count = 0
    for _ in range(int(math.ceil(max_selection_count))):
        flat = int(score_maps.view(-1).argmax())
        i, r, c = flat // (h * w), (flat // w) % h, flat % w
        selected[i].append((r, c, region_size, region_size))
        count += 1
        score_maps[i, max(0, r - region_size):min(h, r + region_size), max(0, c - region_size):min(w, c + region_size)] = 0
        if score_maps.max() < 0.01:
            break
    return selected, count


# ------------------------------------------------------------------ max-subset (max_subset.py:17-39)
def max_representative_samples(image_features, candidate_features, selection_count):
    """greedy facility location: each pick minimises sum_i min(mind_i, D[i, j]) over unpicked candidates j
    (first strict improvement wins), D = sklearn euclidean distances in f64"""
    from sklearn.metrics import pairwise_distances

    dist = pairwise_distances(np.asarray(image_features, dtype=np.float64), np.asarray(candidate_features, dtype=np.float64),
                              metric="euclidean")
    mind = np.full(dist.shape[0], np.inf)
    picked = []
    for _ in range(selection_count):
        best_score, best_j, best_mind = -np.inf, None, None
        for j in range(dist.shape[1]):
            if j in picked:
                continue
            tmp = np.minimum(mind, dist[:, j])
            score = -np.sum(tmp)
            if score > best_score:
                best_score, best_j, best_mind = score, j, tmp
        picked.append(best_j)
        mind = best_mind
    return picked


# ------------------------------------------------------------------ region features (max_subset.py:49-71,91-113)
def region_grid_features(feats, region_size, crop_size):
    """_get_features_for_image_regions: the feature map is cut into a grid of h x w cells (h = floor(region * H / crop)) and
    every cell becomes ONE C-vector.  The reference pools the cell with F.avg_pool2d(cell, (H, W)) -- a kernel of the FULL
    map size over an h x w crop, which PyTorch rejects ("Output size is too small", max_subset.py:62-63 as written can
    never have run); the evident intent, one average per cell, is what is restated here.  -> [B * rows * cols, C] f64"""
    b, c, hh, ww = feats.shape
    h = math.floor(region_size * hh / crop_size)
    w = math.floor(region_size * ww / crop_size)
    rows_, cols_ = math.floor(hh / h), math.floor(ww / w)
    out = []
    for i in range(b):
        for r in range(rows_):
            for q in range(cols_):
                out.append(feats[i, :, r * h:r * h + h, q * w:q * w + w].double().mean(dim=(1, 2)).numpy())
    return np.stack(out) if out else np.zeros((0, c))


def region_features(feats, regions, crop_size):
    """_get_features_for_regions (max_subset.py:91-113): region (r, c, h, w) in image pixels -> feature-map crop
    [floor(r * H / crop) : + floor(h * H / crop)] -> one C-vector (same remark about the pooling kernel)."""
    out = []
    for i, (r0, c0, h0, w0) in enumerate(regions):
        rr, rc = feats.shape[2] / crop_size, feats.shape[3] / crop_size
        r, c, h, w = math.floor(r0 * rr), math.floor(c0 * rc), math.floor(h0 * rr), math.floor(w0 * rc)
        out.append(feats[i, :, r:r + h, c:c + w].double().mean(dim=(1, 2)).numpy())
    return np.stack(out)


# ------------------------------------------------------------------ confusion-matrix metrics (utils/metrics.py:6-49)
def confusion_matrix(gt, pred, num_class):
    """rows = ground truth, columns = prediction, pixels with gt outside [0, num_class) dropped (metrics.py:37-42)"""
    gt = np.asarray(gt)
    pred = np.asarray(pred)
    keep = (gt >= 0) & (gt < num_class)
    idx = gt[keep].astype(np.int64) * num_class + pred[keep].astype(np.int64)
    return np.bincount(idx, minlength=num_class * num_class).reshape(num_class, num_class).astype(np.float64)


def confusion_metrics(cm):
    """-> dict(pixel_acc, class_acc, miou, fwiou) with the reference's nan conventions (metrics.py:13-35)"""
    with np.errstate(divide="ignore", invalid="ignore"):
        diag, rows_, cols_, tot = np.diag(cm), cm.sum(axis=1), cm.sum(axis=0), cm.sum()
        iou = diag / (rows_ + cols_ - diag)
        freq = rows_ / tot
        return dict(pixel_acc=diag.sum() / tot, class_acc=np.nanmean(diag / rows_), miou=np.nanmean(iou),
                    fwiou=(freq[freq > 0] * iou[freq > 0]).sum())
