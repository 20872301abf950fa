"""Restatement of the reference's loss / metric arithmetic in plain torch (CPU, any float dtype). TEST INFRASTRUCTURE ONLY.

  entropic_openset_loss   reference openset_imagenet/losses.py:16-29 (targets built at :17-28, CrossEntropyLoss at :29)
  softmax_loss            reference openset_imagenet/train.py:343    (CrossEntropyLoss(ignore_index=-1))
  garbage_loss            reference openset_imagenet/train.py:344-347 (CrossEntropyLoss(weight=class_weights))
  class_weights           reference openset_imagenet/dataset.py:60-68,77-86 (replace_negative_label + calculate_class_weights)
  confidence              reference openset_imagenet/metrics.py:8-42
  objectosphere_loss      NOT in the reference snapshot: build-defined, parity unpinned.
Pinned against vectors produced by the reference's own files: tests/golden/losses_reference.npz (tests/test_oracle.py).
"""
import torch


def _log_softmax(z):
    return z - torch.logsumexp(z, dim=1, keepdim=True)


def entropic_targets(target, C, unk_weight, dtype):
    t = torch.zeros(target.shape[0], C, dtype=dtype)
    known = target >= 0
    if known.any():
        t[known, target[known]] = 1.0
    t[~known, :] = unk_weight / C
    return t


def entropic_openset_loss(logits, target, unk_weight=1.0):
    """J = -(1/B) sum_i sum_c t_ic log_softmax(z_i)_c ; every negative label (-1, -2, ...) is 'unknown' (losses.py:18)."""
    t = entropic_targets(target, logits.shape[1], unk_weight, logits.dtype)
    return -(t * _log_softmax(logits)).sum(dim=1).mean()


def softmax_loss(logits, target, ignore_index=-1):
    """Mean of -log_softmax(z)_y over rows with y != ignore_index; no such row -> NaN (torch semantics, kept)."""
    keep = target != ignore_index
    ls = _log_softmax(logits)
    picked = ls[keep, target[keep]]
    return -picked.sum() / keep.sum().to(logits.dtype)


def garbage_loss(logits, target, class_weights):
    """-sum_i w_{y_i} log_softmax(z_i)_{y_i} / sum_i w_{y_i}"""
    ls = _log_softmax(logits)
    w = class_weights.to(logits.dtype)[target]
    return -(w * ls[torch.arange(target.shape[0]), target]).sum() / w.sum()


def objectosphere_loss(logits, target, features, unk_weight=1.0, xi=10.0, alpha=1e-4):
    """entropic + alpha/B * sum r_i^2, r_i = max(xi - |f_i|, 0) (known) | |f_i| (unknown). Build-defined."""
    nrm = features.norm(dim=1)
    r = torch.where(target >= 0, (xi - nrm).clamp(min=0), nrm)
    return entropic_openset_loss(logits, target, unk_weight) + alpha * (r * r).mean()


def class_weights(labels):
    """labels: 1-D int tensor of the training CSV's label column (may contain -1). Mirrors dataset.py: -1 is relabelled to
    the largest label + 1, then w_c = N / (count_c * n_labels), ordered by ascending label."""
    labels = labels.clone()
    label_count = torch.unique(labels).numel()      # dataset.py:26 (counts the -1 class)
    labels[labels == -1] = label_count - 1          # dataset.py:65-66
    uniq, counts = torch.unique(labels, return_counts=True)
    return (labels.numel() / (counts.double() * uniq.numel())).float()


def confidence(scores, target_labels, offset=0.0, unknown_class=-1, last_valid_class=None):
    """(kn_conf, kn_count, neg_conf, neg_count): mean confidences and counts, as reference metrics.py:8-42 returns them."""
    with torch.no_grad():
        unknown = target_labels == unknown_class
        known = (~unknown) & (target_labels >= 0)
        kn_count, neg_count = int(known.sum()), int(unknown.sum())
        kn_conf = neg_conf = 0.0
        if kn_count:
            kn_conf = float(scores[known, target_labels[known]].sum()) / kn_count
        if neg_count:
            neg_conf = float((1.0 + offset - scores[unknown, :last_valid_class].max(dim=1)[0]).sum()) / neg_count
    return kn_conf, kn_count, neg_conf, neg_count
