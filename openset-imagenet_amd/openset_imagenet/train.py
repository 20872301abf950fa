"""Training loop, loss / optimizer construction and checkpointing with the call contract of the reference
openset_imagenet/train.py, on top of the MI355X executor.

  set_seeds(seed)                                            reference train.py:23-33
  save_checkpoint(f_name, model, epoch, opt, best_score_, scheduler=None)     train.py:37-60   (same dict schema)
  load_checkpoint(model, checkpoint, opt=None, scheduler=None) -> (start_epoch, best_score)   train.py:63-101
  train(model, data_loader, optimizer, loss_fn, trackers, cfg)                train.py:104-139 (same step order)
  build_loss / build_model / build_optimizer                                  train.py:329-369 (cfg.loss.type, cfg.opt.*)

Per-step order is the reference's: model.train(), zero_grad, H2D, forward, loss, tracker update, backward, step.
The one deliberate difference: the reference calls `j.item()` before `backward()` on every step (train.py:136), a full
device sync per step; here the loss scalars stay on the device and `trackers["j"]` receives exactly the same sequence
of `update(value, batch_len)` calls once, at the end of the epoch (identical avg / sum / count / val).
"""
import pathlib
import random
from collections import OrderedDict

import numpy as np
import torch

from . import dp as _dp
from . import losses as _losses
from . import optim as _optim
from . import tools
from .model import ResNet50


def set_seeds(seed):
    """Seed torch / random / numpy (reference train.py:23-33)."""
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)


def _unwrap(model):
    return model.module if isinstance(model, (_dp.DistributedDataParallel, torch.nn.parallel.DistributedDataParallel)) else model


def save_checkpoint(f_name, model, epoch, opt, best_score_, scheduler=None):
    """Write {"epoch": epoch+1, "model_state_dict", "opt_state_dict", "best_score"[, "scheduler"]} (reference train.py:37-60)."""
    data = {"epoch": epoch + 1,
            "model_state_dict": _unwrap(model).state_dict(),
            "opt_state_dict": opt.state_dict(),
            "best_score": best_score_}
    if scheduler is not None:
        data["scheduler"] = scheduler.state_dict()
    torch.save(data, f_name)


def load_checkpoint(model, checkpoint, opt=None, scheduler=None):
    """Load a checkpoint written by this package or by the reference; strips a DDP "module." prefix
    (reference train.py:63-101). Returns (start_epoch, best_score); raises Exception when the file is missing."""
    file_path = pathlib.Path(checkpoint)
    if not file_path.is_file():
        raise Exception(f"Checkpoint file '{checkpoint}' not found")
    data = torch.load(file_path, map_location=tools.get_device(), weights_only=False)
    state = data["model_state_dict"]
    if list(state.keys())[0][:6] == "module":
        state = OrderedDict((k[7:], v) for k, v in state.items())
    _unwrap(model).load_state_dict(state)
    if opt is not None:
        opt.load_state_dict(data["opt_state_dict"])
    if scheduler is not None:
        scheduler.load_state_dict(data["scheduler"])
    return data["epoch"], data["best_score"]


def build_model(cfg, n_classes):
    """ResNet50(fc_layer_dim=n_classes, out_features=n_classes, logit_bias=False) on the global device (train.py:350-353)."""
    return tools.device(ResNet50(fc_layer_dim=n_classes, out_features=n_classes, logit_bias=False))


def build_loss(cfg, n_classes, class_weights=None):
    """cfg.loss.type in {entropic, softmax, garbage} as in the reference (train.py:339-347); `objectosphere` is this build's
    addition (keys loss.xi, loss.alpha)."""
    kind = cfg.loss.type
    if kind == "entropic":
        return _losses.EntropicOpensetLoss(n_classes, cfg.loss.w)
    if kind == "softmax":
        return _losses.SoftmaxLoss(ignore_index=-1)
    if kind == "garbage":
        if class_weights is None:
            raise ValueError("garbage loss needs the class weights of the training set (dataset.calculate_class_weights)")
        return _losses.GarbageLoss(tools.device(class_weights))
    if kind == "objectosphere":
        return _losses.ObjectosphereLoss(n_classes, cfg.loss.w, getattr(cfg.loss, "xi", 10.0), getattr(cfg.loss, "alpha", 1e-4))
    raise ValueError(f"unknown loss type {kind!r}")


def build_optimizer(cfg, model):
    """Adam(lr) or SGD(lr, momentum=0.9) over the model's arena (train.py:356-359)."""
    if cfg.opt.type == "sgd":
        return _optim.SGD(_unwrap(model), lr=cfg.opt.lr, momentum=0.9)
    return _optim.Adam(_unwrap(model), lr=cfg.opt.lr)


def train(model, data_loader, optimizer, loss_fn, trackers, cfg):
    """One epoch of training (reference train.py:104-139)."""
    for metric in trackers.values():
        metric.reset()
    if not cfg.parallel:
        import tqdm
        data_loader = tqdm.tqdm(data_loader)
    wants_features = isinstance(loss_fn, _losses.ObjectosphereLoss)
    pending, counts = [], []
    for images, labels in data_loader:
        model.train()  # batch-norm uses and collects batch statistics
        batch_len = labels.shape[0]
        optimizer.zero_grad()
        images = tools.device(images)
        labels = tools.device(labels)
        logits, features = model(images)
        j = loss_fn(logits, labels, features) if wants_features else loss_fn(logits, labels)
        pending.append(j.detach())
        counts.append(batch_len)
        j.backward()
        optimizer.step()
    if pending:
        for value, n in zip(torch.stack(pending).cpu().tolist(), counts):
            trackers["j"].update(value, n)


def validate(model, data_loader, loss_fn, n_classes, trackers, cfg):
    """Validation loop with the reference's contract (train.py:142-196): eval-mode forward under no_grad, loss per batch into
    trackers["j"], known / negative confidences into trackers["conf_kn"] / ["conf_unk"].

    The reference fills an [N_val, C] softmax matrix on the device and reduces it with metrics.confidence() in Python loops
    (`sum(known)`, metrics.py:27-28). Here softmax + confidence are one kernel per batch that accumulates the four sums in a
    double[4] on the device (osi_confidence_accumulate); nothing is synchronised until the end of the loop."""
    from . import _native as N
    for metric in trackers.values():
        metric.reset()
    if cfg.loss.type == "garbage":
        min_unk_score, unknown_class, last_valid = 0.0, n_classes - 1, -1
    else:
        min_unk_score, unknown_class, last_valid = 1.0 / n_classes, -1, 0   # 0 encodes Python's None (all columns)
    wants_features = isinstance(loss_fn, _losses.ObjectosphereLoss)
    model.eval()
    losses, counts = [], []
    acc = None
    with torch.no_grad():
        for images, labels in data_loader:
            images = tools.device(images)
            labels = tools.device(labels)
            logits, features = model(images)
            j = loss_fn(logits, labels, features) if wants_features else loss_fn(logits, labels)
            losses.append(j)
            counts.append(labels.shape[0])
            if acc is None:
                acc = torch.zeros(4, dtype=torch.float64, device=logits.device)
            lg = logits.contiguous()
            N.check(N.lib().osi_confidence_accumulate(N.ptr(lg), N.ptr(labels), lg.shape[0], lg.shape[1], float(min_unk_score),
                                                      int(unknown_class), int(last_valid), N.ptr(acc), N.stream_of(lg)),
                    "osi_confidence_accumulate")
    if not losses:
        return
    for value, n in zip(torch.stack(losses).cpu().tolist(), counts):
        trackers["j"].update(value, n)
    kn_sum, kn_count, neg_sum, neg_count = acc.cpu().tolist()
    if kn_count:
        trackers["conf_kn"].update(kn_sum / kn_count, int(kn_count))
    if neg_count:
        trackers["conf_unk"].update(neg_sum / neg_count, int(neg_count))


def get_arrays(model, loader):
    """Targets, logits, deep features and softmax scores of a whole dataset as numpy arrays (reference train.py:200-234);
    everything is gathered on the device and copied to the host once."""
    model.eval()
    t, lg, ft, sc = [], [], [], []
    with torch.no_grad():
        for images, labels in loader:
            labels = tools.device(labels)
            logit, feature = model(tools.device(images))
            t.append(labels)
            lg.append(logit)
            ft.append(feature)
            sc.append(_losses.softmax(logit))
    cat = lambda xs: torch.cat(xs).cpu().numpy()
    return cat(t).astype(np.float32), cat(lg), cat(ft), cat(sc)
