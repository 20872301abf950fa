"""Validation metrics with the call contract of the reference's openset_imagenet/metrics.py, on the GPU.

  confidence(scores, target_labels, offset=0., unknown_class=-1, last_valid_class=None)     metrics.py:8-42
        -> (kn_conf, kn_count, neg_conf, neg_count); one kernel (osi_confidence_from_scores) instead of boolean-mask indexing
           and Python `sum(known)` loops over device tensors. validate() uses the logits form (osi_confidence_accumulate) and
           never builds the [N_val, C] score matrix at all.
  predict_objectosphere(logits, features, threshold)                                        metrics.py:45-62
        -> [B, 2] tensor (predicted class or -1, max softmax score): softmax on the fused kernel, the rest is three tensor ops.
The sklearn AUC wrappers of the reference (metrics.py:65-106) are reporting helpers outside the hot path and are not mirrored.
"""
import torch

from . import _native as N
from . import losses as _losses


def confidence(scores, target_labels, offset=0., unknown_class=-1, last_valid_class=None):
    """Model's confidence on known and negative samples (reference metrics.py:8-42); `scores` are softmax scores [N, C]."""
    N.require_gpu_f32(scores, target_labels)
    s = scores.contiguous().float()
    y = target_labels.contiguous().to(torch.int64)
    if last_valid_class == 0:
        raise ValueError("last_valid_class=0 selects no column (scores[:, :0]); use None for all columns")
    acc = torch.zeros(4, dtype=torch.float64, device=s.device)
    N.check(N.lib().osi_confidence_from_scores(N.ptr(s), N.ptr(y), s.shape[0], s.shape[1], float(offset), int(unknown_class),
                                               0 if last_valid_class is None else int(last_valid_class), N.ptr(acc), N.stream_of(s)),
            "osi_confidence_from_scores")
    ks, kc, ns, nc = acc.cpu().tolist()
    return (ks / kc if kc else 0.0), int(kc), (ns / nc if nc else 0.0), int(nc)


def predict_objectosphere(logits, features, threshold):
    """Predicted class (-1 where |f| * max score < threshold) and score (reference metrics.py:45-62)."""
    scores = _losses.softmax(logits)
    pred_score, pred_class = torch.max(scores, dim=1)
    norms = torch.norm(features, p=2, dim=1)
    pred_class = pred_class.clone()
    pred_class[(norms * pred_score) < threshold] = -1
    return torch.stack((pred_class, pred_score), dim=1)
