"""torch-CPU restatement of the reference model's arithmetic, as pure functions over a reference-format state_dict.

Follows, line by line:
  reference openset_imagenet/model.py:17-26  — torchvision resnet50 body, fc = Linear(2048, fc_layer_dim),
                                               logits = Linear(fc_layer_dim, out_features, bias=logit_bias)
  reference openset_imagenet/model.py:37-39  — features = resnet_base(image); logits = logits(features)
  torchvision.models.resnet (pinned only as torchvision>=0.12.0, requirements.txt:6; NOT under /root/reference):
    ResNet._forward_impl: conv1(7x7/2,p3) -> bn1 -> relu -> maxpool(3,2,1) -> layer1..4 -> avgpool(1) -> flatten -> fc
    Bottleneck.forward (v1.5): conv1x1 -> bn -> relu -> conv3x3(stride) -> bn -> relu -> conv1x1 -> bn -> (+ downsample(x)) -> relu
    downsample = conv1x1(stride) + bn on block 0 of every stage; planes 64/128/256/512, blocks 3/4/6/3, expansion 4.
BatchNorm is torch.nn.functional.batch_norm with momentum 0.1, eps 1e-5 (nn.BatchNorm2d defaults).
Works in float32 or float64 (the fp64 run is the arbiter for the 1e-4 logit tolerance).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

STAGES = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))


def conv_inventory():
    """[(name, cin, cout, k, stride, pad)] in nn.Module registration order (53 convs)."""
    convs = [("resnet_base.conv1", 3, 64, 7, 2, 3)]
    inpl = 64
    for s, (planes, blocks, stride) in enumerate(STAGES):
        for b in range(blocks):
            pre = f"resnet_base.layer{s + 1}.{b}."
            st = stride if b == 0 else 1
            convs.append((pre + "conv1", inpl, planes, 1, 1, 0))
            convs.append((pre + "conv2", planes, planes, 3, st, 1))
            convs.append((pre + "conv3", planes, planes * 4, 1, 1, 0))
            if b == 0:
                convs.append((pre + "downsample.0", inpl, planes * 4, 1, st, 0))
            inpl = planes * 4
    return convs


def bn_name(conv_name):
    if conv_name.endswith("downsample.0"):
        return conv_name[:-1] + "1"
    return conv_name.replace("conv", "bn") if "layer" in conv_name else "resnet_base.bn1"


def init_state(fc_layer_dim, out_features, logit_bias=False, dtype=torch.float32, generator=None):
    """Fresh state_dict with torchvision's initialisation (kaiming-normal fan_out convs, BN 1/0, default nn.Linear)."""
    sd = OrderedDict()
    for name, cin, cout, k, _, _ in conv_inventory():
        std = math.sqrt(2.0 / (cout * k * k))
        sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=generator) * std
        bn = bn_name(name)
        sd[bn + ".weight"] = torch.ones(cout)
        sd[bn + ".bias"] = torch.zeros(cout)
        sd[bn + ".running_mean"] = torch.zeros(cout)
        sd[bn + ".running_var"] = torch.ones(cout)
        sd[bn + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)

    def linear(prefix, fin, fout, bias):
        bound = 1.0 / math.sqrt(fin)
        sd[prefix + ".weight"] = (torch.rand(fout, fin, generator=generator) * 2 - 1) * bound
        if bias:
            sd[prefix + ".bias"] = (torch.rand(fout, generator=generator) * 2 - 1) * bound

    linear("resnet_base.fc", 2048, fc_layer_dim, True)
    linear("logits", fc_layer_dim, out_features, logit_bias)
    ordered = OrderedDict()
    for k in state_keys(logit_bias):
        v = sd[k]
        ordered[k] = v.to(dtype) if v.is_floating_point() else v
    return ordered


def randomize_bn(sd, generator=None):
    """Replace the trivial BatchNorm initialisation (gamma 1, beta 0, running 0/1) by seeded random values so that parity
    checks exercise the affine part and the eval-mode (running statistics) path. In place; returns sd."""
    for k in list(sd):
        if k.endswith("running_mean"):
            pre = k[:-len("running_mean")]
            c = sd[k].numel()
            sd[pre + "weight"] = (0.5 + torch.rand(c, generator=generator)).to(sd[k].dtype)
            sd[pre + "bias"] = (0.2 * torch.randn(c, generator=generator)).to(sd[k].dtype)
            sd[pre + "running_mean"] = (0.1 * torch.randn(c, generator=generator)).to(sd[k].dtype)
            sd[pre + "running_var"] = (0.5 + torch.rand(c, generator=generator)).to(sd[k].dtype)
    return sd


def state_keys(logit_bias=False):
    """The 321 (+1 with logit bias) state_dict keys in torchvision / reference order."""
    keys = []
    for name, *_ in conv_inventory():
        bn = bn_name(name)
        keys.append(name + ".weight")
        keys += [bn + s for s in (".weight", ".bias", ".running_mean", ".running_var", ".num_batches_tracked")]
    # torchvision order inside a Bottleneck is conv1,bn1,conv2,bn2,conv3,bn3,downsample — conv_inventory already emits that
    keys += ["resnet_base.fc.weight", "resnet_base.fc.bias", "logits.weight"]
    if logit_bias:
        keys.append("logits.bias")
    return keys


def _bn(sd, prefix, x, training):
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if training:
        sd[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"], training, 0.1, 1e-5)


def forward(sd, image, training=True, taps=None, gates=None, record_gates=None):
    """(logits, features) of the reference model for NCHW `image`. `taps` (dict) optionally receives intermediate tensors.

    Gate pinning (test infrastructure for the whole-network backward check): the network's only non-smooth operations are its
    49 ReLUs and the max-pool's arg-max, and two correct implementations that differ by fp32 rounding take different branches
    wherever a pre-activation is within rounding of zero — an O(1) change per flipped element that dominates any gradient
    comparison (2e-2 relative). `record_gates` (dict) receives the decisions this run took:
        "relu": list of 49 bool tensors in forward order (stem — taken AFTER the max-pool, i.e. the gate of the pooled value, which
                is all the backward ever sees of the stem ReLU — then bn1 / bn2 / block output of each of the 16 bottlenecks),
        "pool_idx": int64 [B, 64, Hp, Wp] flat arg-max index into the H*W plane (torch.nn.functional.max_pool2d's convention).
    `gates` (same structure) makes the run TAKE the given decisions instead of its own: relu(x) becomes x * gate and the max-pool
    a gather — identical values wherever the decisions agree, and a gradient that is a smooth function of the inputs.
    """
    def tap(k, v):
        if taps is not None:
            taps[k] = v
        return v

    n_relu = [0]

    def relu(v):
        i = n_relu[0]
        n_relu[0] += 1
        if record_gates is not None:
            record_gates.setdefault("relu", []).append((v > 0).detach())
        if gates is None:
            return F.relu(v)
        return v * gates["relu"][i].to(v.dtype)

    x = F.conv2d(image, sd["resnet_base.conv1.weight"], None, 2, 3)
    tap("conv1", x)
    x = _bn(sd, "resnet_base.bn1", x, training)
    # relu then max-pool == max-pool then relu (monotone), and only the pooled element's gate ever reaches the backward
    if gates is None:
        x, idx = F.max_pool2d(x, 3, 2, 1, return_indices=True)
    else:
        idx = gates["pool_idx"]
        x = torch.gather(x.flatten(2), 2, idx.flatten(2)).view(idx.shape)
    if record_gates is not None:
        record_gates["pool_idx"] = idx.detach()
    x = relu(x)
    tap("maxpool", x)
    for s, (planes, blocks, stride) in enumerate(STAGES):
        for b in range(blocks):
            pre = f"resnet_base.layer{s + 1}.{b}."
            st = stride if b == 0 else 1
            identity = x
            out = relu(_bn(sd, pre + "bn1", F.conv2d(x, sd[pre + "conv1.weight"]), training))
            out = relu(_bn(sd, pre + "bn2", F.conv2d(out, sd[pre + "conv2.weight"], None, st, 1), training))
            out = _bn(sd, pre + "bn3", F.conv2d(out, sd[pre + "conv3.weight"]), training)
            if b == 0:
                identity = _bn(sd, pre + "downsample.1", F.conv2d(x, sd[pre + "downsample.0.weight"], None, st), training)
            x = relu(out + identity)
            tap(f"layer{s + 1}.{b}", x)
    x = torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)
    tap("pooled", x)
    features = F.linear(x, sd["resnet_base.fc.weight"], sd["resnet_base.fc.bias"])
    logits = F.linear(features, sd["logits.weight"], sd.get("logits.bias"))
    return logits, features


def gate_disagreements(a, b):
    """(differing ReLU decisions, differing arg-max positions, total ReLU decisions) between two gate records. An arg-max only
    counts where the pooled element's gate is open on either side: behind a closed gate no gradient flows, and an implementation
    that pools AFTER the ReLU (all-zero window: first element wins) legitimately names another element than one that pools before."""
    d = sum(int((x != y).sum()) for x, y in zip(a["relu"], b["relu"]))
    n = sum(x.numel() for x in a["relu"])
    live = a["relu"][0] | b["relu"][0]
    return d, int(((a["pool_idx"] != b["pool_idx"]) & live).sum()), n


def param_keys(sd):
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]


def forward_backward(sd, image, target, loss_fn, training=True, gates=None, record_gates=None):
    """One forward + loss + backward. Returns (logits, features, loss, {param key: grad}); running stats in `sd` are updated.
    `gates` / `record_gates`: see forward()."""
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in param_keys(sd)}
    work = dict(sd)
    work.update(leaves)
    logits, features = forward(work, image, training, gates=gates, record_gates=record_gates)
    for k in sd:  # carry the in-place buffer updates back
        if k.endswith("num_batches_tracked"):
            sd[k] = work[k]
    loss = loss_fn(logits, target, features)
    loss.backward()
    return logits.detach(), features.detach(), loss.detach(), {k: v.grad for k, v in leaves.items()}


def adam_step(sd, grads, state, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam single-tensor arithmetic (reference train.py:356-357, stepped at train.py:139)."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    bc1, bc2 = 1 - betas[0] ** t, 1 - betas[1] ** t
    for k, g in grads.items():
        m = state.setdefault("m." + k, torch.zeros_like(g))
        v = state.setdefault("v." + k, torch.zeros_like(g))
        m.lerp_(g, 1 - betas[0])
        v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        sd[k] = sd[k] - (lr / bc1) * m / denom


def sgd_step(sd, grads, state, lr=1e-3, momentum=0.9):
    """torch.optim.SGD(momentum=0.9) arithmetic (reference train.py:358-359)."""
    for k, g in grads.items():
        if "b." + k not in state:
            state["b." + k] = g.clone()
        else:
            state["b." + k].mul_(momentum).add_(g)
        sd[k] = sd[k] - lr * state["b." + k]
