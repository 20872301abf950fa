"""Generates tests/golden/misc_reference.npz — DEV CONTAINER ONLY (needs /root/reference; never runs on the GPU box).

Pins the remaining host-side pieces of the loop contract to the reference's OWN code, executed in place:
  * metrics.predict_objectosphere              (/root/reference/openset_imagenet/metrics.py:45-62, loaded by path)
  * losses.AverageMeter, losses.EarlyStopping  (/root/reference/openset_imagenet/losses.py:32-94, loaded by path with the
                                                in-memory `vast.tools.device` identity, as tests/golden/make_golden.py does)
  * util.NameSpace / util.load_yaml            (/root/reference/openset_imagenet/util.py:16-34, on a synthetic YAML text; the module's top imports matplotlib,
                                                which is not installed, so the two definitions are taken from the syntax tree at
                                                run time — like make_golden_oscr.py — and compiled in memory with yaml)
Only arrays and strings of inputs / expected outputs are written; no reference source text goes into the repo.
"""
import ast
import os
import sys

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import load_reference  # noqa: E402

REF_UTIL = "/root/reference/openset_imagenet/util.py"
# a synthetic configuration in the reference's key structure (nested mappings, null, on/off, floats in 1.e-3 spelling, strings)
CFG_TEXT = """name: experiment
checkpoint:
gpu:
parallel: off
data:
  imagenet_path: /data/ILSVRC2012/
  train_file: protocols/p{}_train.csv
seed: 42
batch_size: 64
loss:
  type: garbage
  w: 1.
opt:
  type: adam
  lr: 1.e-3
  decay: 0
dist:
  distributed: True
  gpus: 2
  port: "8889"
"""


def util_symbols():
    tree = ast.parse(open(REF_UTIL).read(), REF_UTIL)
    nodes = [n for n in tree.body if (isinstance(n, ast.ClassDef) and n.name == "NameSpace") or
             (isinstance(n, ast.FunctionDef) and n.name == "load_yaml")]
    ns = {"yaml": yaml}
    exec(compile(ast.Module(body=nodes, type_ignores=[]), REF_UTIL, "exec"), ns)
    return ns["NameSpace"], ns["load_yaml"]


def main():
    out = {}
    metrics = load_reference("metrics")
    losses = load_reference("losses")
    g = torch.Generator().manual_seed(77)
    names = []
    for name, B, C, thr, fscale in (("c30", 24, 30, 1.7, 1.0), ("c116_low", 16, 116, 0.45, 0.3), ("c152_high", 12, 152, 3.0, 2.0), ("one", 1, 5, 0.7, 1.0)):
        z = torch.randn(B, C, generator=g) * 2
        f = torch.randn(B, C, generator=g) * fscale
        r = metrics.predict_objectosphere(z.clone(), f.clone(), thr)
        out[f"po.{name}.logits"], out[f"po.{name}.features"] = z.numpy(), f.numpy()
        out[f"po.{name}.threshold"], out[f"po.{name}.result"] = np.float64(thr), r.numpy()
        names.append(name)
    out["po.names"] = np.array(names)

    # AverageMeter: a sequence of (value, count) updates -> (val, avg, sum, count) after every update, and the repr
    rng = np.random.default_rng(5)
    seq = [(float(rng.normal() * 3), int(rng.integers(1, 130))) for _ in range(12)] + [(0.0, 1), (1e-9, 256)]
    m = losses.AverageMeter()
    trace = []
    for v, c in seq:
        m.update(v, c)
        trace.append((m.val, m.avg, m.sum, m.count))
    out["am.updates"], out["am.trace"], out["am.repr"] = np.array(seq, dtype=np.float64), np.array(trace, dtype=np.float64), np.array(repr(m))
    m.reset()
    out["am.after_reset"] = np.array([m.val, m.avg, m.sum, m.count], dtype=np.float64)

    # EarlyStopping: metric sequences in both modes -> (counter, best_score, early_stop) after every call
    for tag, kwargs, loss_mode, vals in (("metric_p3", dict(patience=3), False, [1.0, 1.2, 1.1, 1.15, 1.3, 1.2, 1.25, 1.29, 1.0]),
                                         ("loss_p2_delta", dict(patience=2, delta=0.05), True, [2.0, 1.9, 1.97, 1.8, 1.82, 1.79, 1.9]),
                                         ("metric_p1", dict(patience=1), False, [0.5, 0.4])):
        es = losses.EarlyStopping(**kwargs)
        tr = []
        for v in vals:
            es(v, loss=loss_mode)
            tr.append((es.counter, es.best_score, float(es.early_stop)))
        out[f"es.{tag}.values"], out[f"es.{tag}.trace"] = np.array(vals), np.array(tr, dtype=np.float64)
        out[f"es.{tag}.args"] = np.array([kwargs.get("patience", 100), kwargs.get("delta", 0), float(loss_mode)], dtype=np.float64)
    out["es.names"] = np.array(["metric_p3", "loss_p2_delta", "metric_p1"])

    # NameSpace / load_yaml on a synthetic configuration: dict() and dump() of the loaded object
    import tempfile
    NameSpace, load_yaml = util_symbols()
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as fh:
        fh.write(CFG_TEXT)
    cfg = load_yaml(fh.name)
    os.unlink(fh.name)
    out["ns.yaml_text"] = np.array(CFG_TEXT)
    out["ns.dump"] = np.array(cfg.dump())
    out["ns.dict_repr"] = np.array(repr(cfg.dict()))
    out["ns.loss_type"], out["ns.lr"], out["ns.gpu_is_none"] = np.array(cfg.loss.type), np.float64(cfg.opt.lr), np.bool_(cfg.gpu is None)
    np.savez_compressed(os.path.join(HERE, "misc_reference.npz"), **out)
    print("wrote misc_reference.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
