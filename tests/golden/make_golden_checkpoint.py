"""Generates tests/golden/checkpoint_reference.pth (+ .json) — DEV CONTAINER ONLY (needs /root/reference; never runs on the GPU box).

Pins the checkpoint format to the reference's OWN code. `save_checkpoint()` and `load_checkpoint()` of
/root/reference/openset_imagenet/train.py:36-101 are taken from the syntax tree at run time (the module cannot be imported whole:
vast / loguru / torchvision / tensorboard are absent), compiled in memory with `vast.tools._device = "cpu"` and torch's own
DistributedDataParallel, and used both ways:

  1. the reference's save_checkpoint writes `checkpoint_reference.pth` for the small (logits, features) model of
     make_golden_loop.py after one epoch of the reference's train() with Adam + StepLR — a real file in the reference's format
     (tests/test_loop_contract.py loads it with THIS package's load_checkpoint and continues training from it);
  2. this package's save_checkpoint writes a file for the same objects and the reference's load_checkpoint reads it back here;
     the outcome (epoch, best score, every tensor equal) goes into `checkpoint_reference.json` together with the nested key
     structure of both files, which the test compares again without the reference.

The .pth holds tensors and Python scalars only (a few KB); no reference source text goes into the repo.
"""
import ast
import json
import os
import pathlib
import sys
import tempfile
import types
from collections import OrderedDict

import numpy as np
import torch
import tqdm

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path[:0] = [HERE, ROOT, os.path.join(ROOT, "openset-imagenet_amd")]
from make_golden import load_reference  # noqa: E402
from make_golden_loop import TinyNet, Loader, NS, C, B, batches  # noqa: E402

REF_TRAIN = "/root/reference/openset_imagenet/train.py"


def reference_functions():
    tree = ast.parse(open(REF_TRAIN).read(), REF_TRAIN)
    want = ("save_checkpoint", "load_checkpoint", "train")
    nodes = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert len(nodes) == len(want)
    vast = types.SimpleNamespace(tools=types.SimpleNamespace(_device="cpu"))
    ns = {"torch": torch, "tqdm": tqdm, "pathlib": pathlib, "OrderedDict": OrderedDict, "vast": vast, "device": lambda x: x,
          "DistributedDataParallel": torch.nn.parallel.DistributedDataParallel}
    exec(compile(ast.Module(body=nodes, type_ignores=[]), REF_TRAIN, "exec"), ns)
    return ns["save_checkpoint"], ns["load_checkpoint"], ns["train"]


def structure(obj):
    """Nested key / type / shape skeleton of a checkpoint dict (what a loader depends on)."""
    if isinstance(obj, torch.Tensor):
        return ["tensor", str(obj.dtype), list(obj.shape)]
    if isinstance(obj, dict):
        return {str(k): structure(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [structure(v) for v in obj]
    return type(obj).__name__


def objects():
    torch.manual_seed(5)
    model = TinyNet(C)
    opt = torch.optim.Adam(params=model.parameters(), lr=1e-2)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)   # lr halves after the saved epoch: a resume that loses the scheduler shows
    return model, opt, sched


def main():
    from openset_imagenet import tools
    from openset_imagenet import train as ours
    tools.set_device_cpu()
    ref_save, ref_load, ref_train = reference_functions()
    losses = load_reference("losses")
    g = torch.Generator().manual_seed(21)
    tr = batches(g, -1, C)
    model, opt, sched = objects()
    init = {k: v.clone() for k, v in model.state_dict().items()}
    ref_train(model, Loader(tr), opt, losses.EntropicOpensetLoss(C, 1.0), {"j": losses.AverageMeter()}, NS(parallel=True))
    sched.step()
    path = os.path.join(HERE, "checkpoint_reference.pth")
    ref_save(path, model, 0, opt, 1.375, sched)                       # 1. the reference writes

    with tempfile.TemporaryDirectory() as d:                          # 2. this package writes, the reference reads
        f = os.path.join(d, "ours.pth")
        ours.save_checkpoint(f, model, 0, opt, 1.375, sched)
        m2, o2, s2 = objects()
        epoch, best = ref_load(m2, f, o2, s2)
        same = all(torch.equal(a, b) for a, b in zip(m2.state_dict().values(), model.state_dict().values()))
        same_opt = all(torch.equal(o2.state[p2][k], opt.state[p][k]) for p, p2 in zip(model.parameters(), m2.parameters())
                       for k in ("exp_avg", "exp_avg_sq"))
        ours_struct = structure(torch.load(f, weights_only=False))
    ref_struct = structure(torch.load(path, weights_only=False))
    # the run goes on for one more epoch from the saved state: what a correct resume must reproduce
    t2 = {"j": losses.AverageMeter()}
    ref_train(model, Loader(tr), opt, losses.EntropicOpensetLoss(C, 1.0), t2, NS(parallel=True))
    after = {k: v.clone() for k, v in model.state_dict().items()}
    meta = {"reference_reads_our_file": {"epoch": epoch, "best_score": best, "model_equal": bool(same), "optimizer_equal": bool(same_opt),
                                         "scheduler_last_epoch": s2.last_epoch},
            "structure_reference_file": ref_struct, "structure_our_file": ours_struct,
            "train_batches_seed": 21, "model_seed": 5}
    assert same and same_opt and (epoch, best) == (1, 1.375) and ours_struct == ref_struct
    json.dump(meta, open(os.path.join(HERE, "checkpoint_reference.json"), "w"), indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "checkpoint_reference_inputs.npz"),
                        **{f"init.{k}": v.numpy() for k, v in init.items()}, **{f"after.{k}": v.numpy() for k, v in after.items()},
                        after_train_j=np.array([t2["j"].val, t2["j"].avg, t2["j"].sum, t2["j"].count]), lr_after_resume=np.float64(opt.param_groups[0]["lr"]),
                        **{f"x{i}": x.numpy() for i, (x, _) in enumerate(tr)}, **{f"y{i}": y.numpy() for i, (_, y) in enumerate(tr)})
    print(path, os.path.getsize(path), "bytes; reference reads our file:", meta["reference_reads_our_file"])


if __name__ == "__main__":
    main()
