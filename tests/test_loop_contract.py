"""The loop contract against the reference's OWN loops: tests/golden/loop_reference.npz holds what `train()`, `validate()` and
`get_arrays()` of reference openset_imagenet/train.py:104-234 produced (executed in place by tests/golden/make_golden_loop.py) on
a small (logits, features) model — tracker values, parameters after each epoch, validation confidences, the gathered arrays.

CPU: this package's train() is model-agnostic above the kernels; with the same torch model, torch.nn.CrossEntropyLoss and torch
optimizer it must reproduce the reference's trackers and parameters exactly (same operations in the same order).
GPU: the same loops with this package's HIP losses, confidence kernel and softmax, within fp32 tolerance of the CPU reference.
"""
import os

import numpy as np
import pytest
import torch

from openset_imagenet import losses as L, tools
from openset_imagenet.train import get_arrays, train, validate
from openset_imagenet.util import NameSpace

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loop_reference.npz")


class TinyNet(torch.nn.Module):
    """The model make_golden_loop.py ran the reference loops on: forward -> (logits, features), `.logits` is a Linear."""

    def __init__(self, hw, feat, n_out):
        super().__init__()
        self.body = torch.nn.Linear(3 * hw * hw, feat, bias=False)
        self.bn = torch.nn.BatchNorm1d(feat)
        self.logits = torch.nn.Linear(feat, n_out)

    def forward(self, x):
        f = torch.relu(self.bn(self.body(x.flatten(1))))
        return self.logits(f), f


class Loader(list):
    def __init__(self, batches):
        super().__init__(batches)
        self.dataset = range(sum(int(y.shape[0]) for _, y in batches))


def _case(g, name):
    C, F, HW, B = (int(v) for v in g["dims"])
    n = len(g["sizes"])
    model = TinyNet(HW, F, C)
    model.load_state_dict({k[len(name) + 6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(name + ".init.")})
    tr = [(torch.from_numpy(g[f"{name}.train.x{i}"]), torch.from_numpy(g[f"{name}.train.y{i}"])) for i in range(n)]
    va = [(torch.from_numpy(g[f"{name}.val.x{i}"]), torch.from_numpy(g[f"{name}.val.y{i}"])) for i in range(n)]
    loss_type, opt_type, epochs = (str(v) for v in g[f"{name}.meta"])
    cfg = NameSpace({"parallel": True, "batch_size": B, "loss": {"type": loss_type}})
    return model, tr, va, loss_type, opt_type, int(epochs), cfg, C


def _optimizer(opt_type, model):
    if opt_type == "adam":
        return torch.optim.Adam(params=model.parameters(), lr=1e-2)
    return torch.optim.SGD(params=model.parameters(), lr=1e-2, momentum=0.9)


def _meter(m):
    return np.array([m.val, m.avg, m.sum, m.count], dtype=np.float64)


@pytest.mark.parametrize("name", ["softmax_sgd", "garbage_adam"])
def test_train_loop_reproduces_the_reference_loop_exactly(name):
    g = np.load(GOLD)
    model, tr, _, loss_type, opt_type, epochs, cfg, C = _case(g, name)
    tools.set_device_cpu()
    loss_fn = torch.nn.CrossEntropyLoss(ignore_index=-1) if loss_type == "softmax" else \
        torch.nn.CrossEntropyLoss(weight=torch.from_numpy(g[f"{name}.class_weights"]))       # reference train.py:343-347
    opt = _optimizer(opt_type, model)
    trackers = {"j": L.AverageMeter()}
    trackers["j"].update(123.0, 7)            # train() resets the trackers first (train.py:115-116)
    for e in range(epochs):
        train(model, Loader(tr), opt, loss_fn, trackers, cfg)
        assert model.training                  # train.py:125
        np.testing.assert_array_equal(_meter(trackers["j"]), g[f"{name}.epoch{e}.train_j"])
        for k, v in model.state_dict().items():
            np.testing.assert_array_equal(v.numpy(), g[f"{name}.epoch{e}.state.{k}"], err_msg=f"{name} epoch {e} {k}")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["entropic_adam", "softmax_sgd", "garbage_adam"])
def test_loops_with_hip_losses_match_the_reference_loops(cuda, name):
    g = np.load(GOLD)
    model, tr, va, loss_type, opt_type, epochs, cfg, C = _case(g, name)
    tools.set_device_gpu(0)
    model = tools.device(model)
    if loss_type == "entropic":
        loss_fn = L.EntropicOpensetLoss(C, 1.0)
    elif loss_type == "softmax":
        loss_fn = L.SoftmaxLoss(ignore_index=-1)
    else:
        loss_fn = L.GarbageLoss(torch.from_numpy(g[f"{name}.class_weights"]).cuda())
    opt = _optimizer(opt_type, model)
    t_tr = {"j": L.AverageMeter()}
    t_va = {"j": L.AverageMeter(), "conf_kn": L.AverageMeter(), "conf_unk": L.AverageMeter()}
    tol = dict(rtol=1e-3, atol=1e-4)          # fp32 on two devices through up to ten Adam steps at lr 1e-2; a contract slip (mode, reset, counts) is orders larger
    for e in range(epochs):
        train(model, Loader(tr), opt, loss_fn, t_tr, cfg)
        np.testing.assert_allclose(_meter(t_tr["j"]), g[f"{name}.epoch{e}.train_j"], **tol)
        validate(model, Loader(va), loss_fn, C, t_va, cfg)
        assert not model.training              # train.py:166
        for k in t_va:
            np.testing.assert_allclose(_meter(t_va[k]), g[f"{name}.epoch{e}.val_{k}"], err_msg=f"{name} epoch {e} {k}", **tol)
        for k, v in model.state_dict().items():
            np.testing.assert_allclose(v.cpu().numpy(), g[f"{name}.epoch{e}.state.{k}"], err_msg=f"{name} epoch {e} {k}", **tol)
    arrays = get_arrays(model, Loader(va))
    for k, a in zip(("targets", "logits", "features", "scores"), arrays):
        ref = g[f"{name}.arrays.{k}"]
        assert a.shape == ref.shape and a.dtype == ref.dtype, k
        np.testing.assert_allclose(a, ref, err_msg=f"{name} arrays {k}", **tol)


# ---- checkpoints: a file written by the reference's own save_checkpoint (tests/golden/make_golden_checkpoint.py) --------------------
def _structure(obj):
    if isinstance(obj, torch.Tensor):
        return ["tensor", str(obj.dtype), list(obj.shape)]
    if isinstance(obj, dict):
        return {str(k): _structure(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_structure(v) for v in obj]
    return type(obj).__name__


def test_resume_from_a_checkpoint_written_by_the_reference(tmp_path):
    import json
    from openset_imagenet.train import load_checkpoint, save_checkpoint
    from oracle import losses_oracle as LO
    gdir = os.path.dirname(GOLD)
    meta = json.load(open(os.path.join(gdir, "checkpoint_reference.json")))
    inp = np.load(os.path.join(gdir, "checkpoint_reference_inputs.npz"))
    g = np.load(GOLD)
    C, F, HW, B = (int(v) for v in g["dims"])
    tools.set_device_cpu()
    model = TinyNet(HW, F, C)
    opt = torch.optim.Adam(params=model.parameters(), lr=1e-2)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)
    epoch, best = load_checkpoint(model, os.path.join(gdir, "checkpoint_reference.pth"), opt, sched)      # reference train.py:63-101
    assert (epoch, best) == (1, 1.375) and sched.last_epoch == 1
    assert opt.param_groups[0]["lr"] == float(inp["lr_after_resume"]) == 5e-3
    # this package's save_checkpoint writes the same nested structure the reference's wrote, and (recorded at generation time) the
    # reference's load_checkpoint read such a file back with every tensor equal
    f = tmp_path / "ours.pth"
    save_checkpoint(f, model, 0, opt, 1.375, sched)
    assert _structure(torch.load(f, weights_only=False)) == meta["structure_reference_file"] == meta["structure_our_file"]
    assert meta["reference_reads_our_file"] == {"epoch": 1, "best_score": 1.375, "model_equal": True, "optimizer_equal": True,
                                                "scheduler_last_epoch": 1}
    # the resumed run continues exactly where the reference's own run went (same batches, entropic loss from the oracle)
    n = len(g["sizes"])
    tr = [(torch.from_numpy(inp[f"x{i}"]), torch.from_numpy(inp[f"y{i}"])) for i in range(n)]
    trackers = {"j": L.AverageMeter()}
    train(model, Loader(tr), opt, lambda z, y: LO.entropic_openset_loss(z, y, 1.0), trackers, NameSpace({"parallel": True}))
    np.testing.assert_allclose(_meter(trackers["j"]), inp["after_train_j"], rtol=1e-6)
    for k, v in model.state_dict().items():
        np.testing.assert_allclose(v.numpy(), inp[f"after.{k}"], rtol=1e-5, atol=1e-7, err_msg=k)
