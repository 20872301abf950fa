"""Generates tests/golden/loop_reference.npz — DEV CONTAINER ONLY (needs /root/reference; never runs on the GPU box).

Pins the loop contract to the reference's OWN code: `train()`, `validate()` and `get_arrays()` of
/root/reference/openset_imagenet/train.py:104-234 are taken from the syntax tree at run time (the module's top imports — vast,
loguru, torchvision, tensorboard — are not installed, so it cannot be imported whole), compiled in memory and run on the CPU with
  device      -> identity (what vast.tools.device is without a GPU)
  confidence  -> the reference's metrics.confidence, loaded by path
  loss_fn     -> the reference's EntropicOpensetLoss (losses.py, loaded by path) or torch.nn.CrossEntropyLoss (train.py:343-347)
  optimizer   -> torch.optim.Adam / SGD(momentum=0.9) as train.py:356-359 builds them
on a small two-layer torch model that returns (logits, features) like the reference's ResNet50 wrapper (model.py:44-47). The
fixture holds the model's initial weights, the batches, and what the reference's loops produced: tracker values, parameters after
the epoch, validation trackers, get_arrays() outputs. Only arrays go into the repo; no reference source text.
"""
import ast
import os
import sys

import numpy as np
import torch
import tqdm

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference  # noqa: E402

REF_TRAIN = "/root/reference/openset_imagenet/train.py"
C, F, HW, B = 6, 10, 8, 6
SIZES = (6, 6, 6, 6, 3)       # a ragged last batch, as every real epoch has


class TinyNet(torch.nn.Module):
    """Stands where the reference's ResNet50 wrapper stands in the loops: forward -> (logits, features), `.logits` is a Linear."""

    def __init__(self, n_out):
        super().__init__()
        self.body = torch.nn.Linear(3 * HW * HW, F, bias=False)   # like the convolutions in front of the real batch norms:
        # a bias there has a mathematically zero gradient, and Adam would turn its rounding noise into lr-sized steps
        self.bn = torch.nn.BatchNorm1d(F)          # train()/eval() must matter, as with the real network
        self.logits = torch.nn.Linear(F, n_out)

    def forward(self, x):
        f = torch.relu(self.bn(self.body(x.flatten(1))))
        return self.logits(f), f


class Loader(list):
    """A list of (images, labels) batches with the `.dataset` the reference's validate()/get_arrays() take the length of."""

    def __init__(self, batches):
        super().__init__(batches)
        self.dataset = range(sum(int(y.shape[0]) for _, y in batches))


def reference_loops():
    tree = ast.parse(open(REF_TRAIN).read(), REF_TRAIN)
    nodes = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("train", "validate", "get_arrays")]
    assert len(nodes) == 3
    ns = {"torch": torch, "tqdm": tqdm, "device": lambda x: x, "confidence": load_reference("metrics").confidence}
    exec(compile(ast.Module(body=nodes, type_ignores=[]), REF_TRAIN, "exec"), ns)
    return ns["train"], ns["validate"], ns["get_arrays"]


class NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def batches(g, labels_lo, n_out):
    out = []
    for b in SIZES:
        x = torch.rand(b, 3, HW, HW, generator=g)
        y = torch.randint(labels_lo, n_out, (b,), generator=g)
        out.append((x, y))
    return out


def meter_tuple(m):
    return np.array([m.val, m.avg, m.sum, m.count], dtype=np.float64)


def main():
    ref_train, ref_validate, ref_get_arrays = reference_loops()
    losses = load_reference("losses")
    out = {"dims": np.array([C, F, HW, B]), "sizes": np.array(SIZES)}
    cases = []
    for name, loss_type, opt_type, labels_lo, epochs in (("entropic_adam", "entropic", "adam", -1, 2), ("softmax_sgd", "softmax", "sgd", 0, 2),
                                                         ("garbage_adam", "garbage", "adam", 0, 1)):
        g = torch.Generator().manual_seed(len(cases) + 11)
        torch.manual_seed(len(cases) + 3)
        model = TinyNet(C)
        for k, v in model.state_dict().items():
            out[f"{name}.init.{k}"] = v.clone().numpy()
        tr, va = batches(g, labels_lo, C), batches(g, -1, C)   # validation keeps the negatives for every loss (train.py:291: only the softmax TRAINING set drops them)
        if loss_type == "garbage":      # the background class is the last index in training; validation keeps -1 -> replaced as dataset.py does
            va = [(x, torch.where(y < 0, torch.tensor(C - 1), y)) for x, y in va]
        for i, (x, y) in enumerate(tr):
            out[f"{name}.train.x{i}"], out[f"{name}.train.y{i}"] = x.numpy(), y.numpy()
        for i, (x, y) in enumerate(va):
            out[f"{name}.val.x{i}"], out[f"{name}.val.y{i}"] = x.numpy(), y.numpy()
        if loss_type == "entropic":
            loss_fn = losses.EntropicOpensetLoss(C, 1.0)
        elif loss_type == "softmax":
            loss_fn = torch.nn.CrossEntropyLoss(ignore_index=-1)
        else:
            w = torch.linspace(0.5, 1.5, C)
            out[f"{name}.class_weights"] = w.numpy()
            loss_fn = torch.nn.CrossEntropyLoss(weight=w)
        opt = torch.optim.Adam(params=model.parameters(), lr=1e-2) if opt_type == "adam" else \
            torch.optim.SGD(params=model.parameters(), lr=1e-2, momentum=0.9)
        cfg = NS(parallel=True, batch_size=B, loss=NS(type=loss_type))
        t_tr = {"j": losses.AverageMeter()}
        t_va = {"j": losses.AverageMeter(), "conf_kn": losses.AverageMeter(), "conf_unk": losses.AverageMeter()}
        for e in range(epochs):
            ref_train(model, Loader(tr), opt, loss_fn, t_tr, cfg)
            out[f"{name}.epoch{e}.train_j"] = meter_tuple(t_tr["j"])
            ref_validate(model, Loader(va), loss_fn, C, t_va, cfg)
            for k in t_va:
                out[f"{name}.epoch{e}.val_{k}"] = meter_tuple(t_va[k])
            for k, v in model.state_dict().items():
                out[f"{name}.epoch{e}.state.{k}"] = v.clone().numpy()
        arrays = ref_get_arrays(model, Loader(va))
        for k, a in zip(("targets", "logits", "features", "scores"), arrays):
            out[f"{name}.arrays.{k}"] = a
        out[f"{name}.meta"] = np.array([loss_type, opt_type, str(epochs)])
        cases.append(name)
    out["cases"] = np.array(cases)
    path = os.path.join(HERE, "loop_reference.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", len(out), "arrays")
    for n in cases:
        print(n, out[f"{n}.epoch0.train_j"], out[f"{n}.epoch0.val_conf_kn"], out[f"{n}.epoch0.val_conf_unk"])


if __name__ == "__main__":
    main()
