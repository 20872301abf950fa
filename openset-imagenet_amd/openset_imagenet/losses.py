"""Open-set training losses as single fused HIP launches, plus the host-side meters of the reference loop.

Call contract of the reference (openset_imagenet/losses.py:7-29 and train.py:341-347): a loss object is called as
`j = loss_fn(logits[B,C] float32, target[B] int64)` and returns a 0-dim tensor supporting `.item()` and `.backward()`.

  EntropicOpensetLoss(num_of_classes, unk_weight=1)   reference losses.py:9-29
  SoftmaxLoss(ignore_index=-1)                         = torch.nn.CrossEntropyLoss(ignore_index=-1),   train.py:343
  GarbageLoss(class_weights)                           = torch.nn.CrossEntropyLoss(weight=class_weights), train.py:344-347
  ObjectosphereLoss(num_of_classes, unk_weight, xi, alpha)  NOT in the reference snapshot (only metrics.predict_objectosphere,
        metrics.py:45-62): entropic loss + alpha/B * sum_i r_i^2 on the deep features, r_i = max(xi - |f_i|, 0) for knowns and
        |f_i| for unknowns (Dhamija et al. 2018). Called as loss_fn(logits, target, features). Parity: unpinned by the reference.

Each call is ONE kernel (osi_loss_fwd_bwd): log-softmax, the loss value and dJ/dlogits in one pass, no [B,C] target
matrix, no `torch.sum(unk_idx).item()` host sync (reference losses.py:26). AverageMeter / EarlyStopping are host-side
bookkeeping with the reference's semantics (losses.py:32-94).
"""
import torch

from . import _native as N


class _FusedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, features, cfg):
        mode, unk_w, ignore_index, class_w, xi, alpha = cfg
        N.require_gpu_f32(logits, target, features, class_w)
        if logits.dim() != 2 or target.dim() != 1 or target.shape[0] != logits.shape[0]:
            raise ValueError("expected logits [B, C] and target [B]")
        if target.dtype != torch.int64:
            raise TypeError("target must be int64")
        logits_c = logits.contiguous().float()
        target_c = target.contiguous()
        need_grad = ctx.needs_input_grad[0] or ctx.needs_input_grad[2]
        feats_c = None if features is None else features.contiguous().float()
        loss, dlogits, dfeat = N.ops().loss_fwd_bwd(mode, logits_c, target_c, float(unk_w), int(ignore_index), class_w, feats_c,
                                                    float(xi), float(alpha), bool(need_grad))
        dlogits = dlogits if need_grad else None
        dfeat = dfeat if features is not None else None
        ctx.save_for_backward(dlogits if dlogits is not None else loss.new_empty(0),
                              dfeat if dfeat is not None else loss.new_empty(0))
        ctx.has_feat = features is not None
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        dlogits, dfeat = ctx.saved_tensors
        gl = dlogits * grad_out if dlogits.numel() else None
        gf = dfeat * grad_out if ctx.has_feat and dfeat.numel() else None
        return gl, None, gf, None


class _LossValue(torch.Tensor):
    """The 0-dim loss a fused loss returns when its logits come straight from the MI355X ResNet50. It is an ordinary tensor
    (`.item()`, arithmetic, autograd all work); only the plain `j.backward()` of the reference loop (train.py:138) is special:
    dJ/dlogits (and dJ/dfeatures) — already computed by the loss kernel — go straight to the network's backward instead of
    through autograd's seed gradient (a ones_like fill) and a `dlogits * 1` multiply: two elementwise launches fewer between the
    forward and the backward pass. Every other use (gradient=, inputs=, retain_graph, create_graph, or a loss that was combined
    with other terms and therefore is a new tensor) takes the ordinary autograd route, which gives the same gradients.

    Divergence from `torch.Tensor.backward`, by design: the direct route does not run autograd, so it is taken only when nothing could
    observe the difference — logits / features carry no tensor hooks and do not retain their gradient (otherwise: the autograd route) —
    and it consumes the loss: a second plain `backward()` of the same loss raises a clear error, as torch's own "backward through the
    graph a second time" would."""

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        if self.__dict__.get("_osi_consumed"):
            raise RuntimeError("backward() was already called on this loss: the network's activations of that forward have been consumed "
                               "(run the forward pass again — the reference loop does, train.py:132-139)")
        direct = self.__dict__.pop("_osi_direct", None)
        if direct is not None:
            watched = any(t is not None and (t.retains_grad or bool(getattr(t, "_backward_hooks", None))) for t in direct[2])
            if watched:
                direct = None
        if direct is None or gradient is not None or retain_graph or create_graph or inputs is not None:
            return super().backward(gradient, retain_graph, create_graph, inputs)
        node, fn, _ = direct
        self.__dict__["_osi_consumed"] = True
        model = node.model
        if node.serial != model._fwd_serial:
            raise RuntimeError("backward() of a forward pass that is no longer the model's latest one: the executor keeps the "
                               "activations of ONE forward (the reference loop is forward, loss, backward, step — train.py:132-139)")
        dlogits, dfeat = fn.saved_tensors
        model._run_backward(dlogits if dlogits.numel() else None, dfeat if (fn.has_feat and dfeat.numel()) else None)


def _loss_value(j, logits, features=None):
    """Wrap the loss for the direct backward when `logits` (and `features`) are the two outputs of one ResNet50 forward."""
    node = logits.grad_fn
    if not j.requires_grad or node is None or getattr(node, "model", None) is None or not hasattr(node, "serial"):
        return j
    if features is not None and features.grad_fn is not node:
        return j
    out = j.as_subclass(_LossValue)
    out._osi_direct = (node, j.grad_fn, (logits, features))
    return out


class EntropicOpensetLoss:
    """Entropic open-set loss (reference losses.py:7-29): one-hot targets for y >= 0, w/C for every negative label,
    mean over all rows of the batch."""

    def __init__(self, num_of_classes, unk_weight=1):
        self.class_count = int(num_of_classes)
        self.unk_weight = float(unk_weight)
        self.unknowns_multiplier = self.unk_weight / self.class_count

    def __call__(self, logits, target):
        if logits.shape[1] != self.class_count:
            raise ValueError(f"logits have {logits.shape[1]} classes, loss was built for {self.class_count}")
        return _loss_value(_FusedLoss.apply(logits, target, None, (N.LOSS_ENTROPIC, self.unk_weight, -1, None, 0.0, 0.0)), logits)


class SoftmaxLoss:
    """torch.nn.CrossEntropyLoss(ignore_index=-1) of reference train.py:343."""

    def __init__(self, ignore_index=-1):
        self.ignore_index = int(ignore_index)

    def __call__(self, logits, target):
        return _loss_value(_FusedLoss.apply(logits, target, None, (N.LOSS_SOFTMAX, 1.0, self.ignore_index, None, 0.0, 0.0)), logits)


class GarbageLoss:
    """torch.nn.CrossEntropyLoss(weight=class_weights) of reference train.py:344-347 (background class = last index)."""

    def __init__(self, class_weights):
        self.weight = class_weights.detach().float().contiguous()

    def __call__(self, logits, target):
        if self.weight.device != logits.device:
            self.weight = self.weight.to(logits.device)
        if self.weight.numel() != logits.shape[1]:
            raise ValueError("class_weights length must equal the number of logits")
        return _loss_value(_FusedLoss.apply(logits, target, None, (N.LOSS_GARBAGE, 1.0, -100, self.weight, 0.0, 0.0)), logits)


class ObjectosphereLoss:
    """Entropic open-set loss + feature-magnitude term (build-defined, see module docstring)."""

    def __init__(self, num_of_classes, unk_weight=1, xi=10.0, alpha=1e-4):
        self.class_count = int(num_of_classes)
        self.unk_weight, self.xi, self.alpha = float(unk_weight), float(xi), float(alpha)

    def __call__(self, logits, target, features):
        return _loss_value(_FusedLoss.apply(logits, target, features,
                                            (N.LOSS_ENTROPIC, self.unk_weight, -1, None, self.xi, self.alpha)), logits, features)


def softmax(logits):
    """Row softmax on the GPU (validation path, reference train.py:177)."""
    N.require_gpu_f32(logits)
    return N.ops().softmax(logits.contiguous().float())


class AverageMeter:
    """Running sample-weighted mean (reference losses.py:32-60)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val, self.avg, self.sum, self.count = 0, 0, 0, 0

    def update(self, val, count=1):
        self.val = val
        self.sum += val * count
        self.count += count
        self.avg = self.sum / self.count

    def __repr__(self):
        return f"{self.avg:3.3f}"


class EarlyStopping:
    """Patience counter on a validation metric (reference losses.py:65-94)."""

    def __init__(self, patience=100, delta=0):
        self.patience, self.delta = patience, delta
        self.counter, self.best_score, self.early_stop = 0, None, False

    def __call__(self, metrics, loss=True):
        score = -metrics if loss is True else metrics
        if self.best_score is None:
            self.best_score = score
        elif score < self.best_score + self.delta:
            self.counter += 1
            if self.counter >= self.patience:
                self.early_stop = True
        else:
            self.best_score, self.counter = score, 0
