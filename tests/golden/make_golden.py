"""Generates the golden vectors under tests/golden/ — DEV CONTAINER ONLY (needs /root/reference; never runs on the GPU box).

Part 1 (pins the oracle to the reference): executes the reference's OWN files
    /root/reference/openset_imagenet/losses.py   (EntropicOpensetLoss)
    /root/reference/openset_imagenet/metrics.py  (confidence)
    /root/reference/openset_imagenet/dataset.py  (replace_negative_label, calculate_class_weights)
loaded by file path (the package __init__ cannot be imported: torchvision / vast / loguru / tensorboard / robustness are not
installed and cannot be fetched). The one third-party symbol losses.py touches, `vast.tools.device`, is an identity on CPU and
is provided in-memory as such. torch.nn.CrossEntropyLoss — which train.py:343-347 instantiates for the softmax
and garbage losses — is called directly with the reference's arguments.
Part 2 (anchors the model restatement): small whole-model vectors from oracle/resnet50_oracle.py at fixed seeds, fp32 and fp64.

Only arrays (inputs / expected outputs) are written; no reference source text goes into the repo.
"""
import importlib.util
import io
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
REF = "/root/reference/openset_imagenet"
sys.path.insert(0, ROOT)


def load_reference(name):
    if "vast" not in sys.modules:
        vast = types.ModuleType("vast")
        tools = types.ModuleType("vast.tools")
        tools.device = lambda x: x  # vast.tools.device on CPU
        vast.tools = tools
        sys.modules["vast"], sys.modules["vast.tools"] = vast, tools
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def loss_cases():
    """(name, B, C, labels builder, unk_weight) — every edge case of SURVEY.md Appendix B."""
    g = torch.Generator().manual_seed(1234)
    cases = []

    def mk(name, B, C, labels, w=1.0, scale=3.0):
        z = torch.randn(B, C, generator=g) * scale
        cases.append((name, z, torch.tensor(labels, dtype=torch.int64), w))

    mk("mixed_c30", 16, 30, [0, 5, -1, 29, -1, 3, 3, -1, 7, 11, -1, 2, 28, -1, -1, 1])
    mk("mixed_w05", 12, 30, [0, -1, -1, 29, 4, -1, 9, 10, -1, 2, 2, -1], w=0.5)
    mk("mixed_w2", 12, 116, [115, -1, 0, -1, 57, 57, -1, 3, 100, -1, -1, 8], w=2.0)
    mk("label_minus2", 8, 30, [0, -2, -1, 4, -2, 29, 1, -2])
    mk("all_known", 8, 151, [150, 0, 1, 2, 75, 75, 149, 33])
    mk("all_unknown", 8, 30, [-1] * 8)
    mk("single_row", 1, 30, [7])
    mk("single_unknown", 1, 116, [-1])
    mk("big_logits", 6, 30, [0, -1, 3, -1, 29, 5], scale=40.0)
    mk("c152_b33", 33, 152, [(i * 37) % 152 if i % 4 else -1 for i in range(33)])
    return cases


def main():
    ref_losses = load_reference("losses")
    ref_metrics = load_reference("metrics")
    ref_dataset = load_reference("dataset")
    out = {}
    # ---- entropic open-set loss (reference losses.py) -------------------------------------------------------
    names = []
    for name, z, y, w in loss_cases():
        zz = z.clone().requires_grad_(True)
        j = ref_losses.EntropicOpensetLoss(num_of_classes=z.shape[1], unk_weight=w)(zz, y)
        j.backward()
        out[f"eos.{name}.logits"], out[f"eos.{name}.target"] = z.numpy(), y.numpy()
        out[f"eos.{name}.w"] = np.float32(w)
        out[f"eos.{name}.loss"], out[f"eos.{name}.dlogits"] = j.detach().numpy(), zz.grad.numpy()
        names.append(name)
    out["eos.names"] = np.array(names)
    # ---- softmax loss: CrossEntropyLoss(ignore_index=-1) (reference train.py:343) --------------------------
    names = []
    for name, z, y, w in loss_cases():
        y2 = y.clone()
        y2[y2 < 0] = -1  # validation data only holds -1 (train.py:291-293 drops negatives from the training set)
        zz = z.clone().requires_grad_(True)
        j = torch.nn.CrossEntropyLoss(ignore_index=-1)(zz, y2)
        j.backward()
        out[f"sm.{name}.logits"], out[f"sm.{name}.target"] = z.numpy(), y2.numpy()
        out[f"sm.{name}.loss"], out[f"sm.{name}.dlogits"] = j.detach().numpy(), zz.grad.numpy()
        names.append(name)
    out["sm.names"] = np.array(names)
    # ---- garbage loss: replace_negative_label + calculate_class_weights + CrossEntropyLoss(weight) ----------
    names = []
    for name, z, y, w in loss_cases():
        C = z.shape[1]
        # a synthetic training CSV whose label histogram defines the class weights: classes 0..C-2 known, -1 negatives
        rng = np.random.RandomState(len(name) * 7 + C)
        counts = rng.randint(3, 40, size=C)  # last entry = number of -1 samples
        labels = np.concatenate([np.full(counts[c], c if c < C - 1 else -1) for c in range(C)])
        with tempfile.NamedTemporaryFile("w", suffix=".csv", delete=False) as f:
            for i, l in enumerate(labels):
                f.write(f"train/x/{i}.JPEG,{l}\n")
            path = f.name
        ds = ref_dataset.ImagenetDataset(path, "/nonexistent")
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ds.replace_negative_label()
        cw = ds.calculate_class_weights()
        os.unlink(path)
        y2 = y.clone()
        y2[y2 < 0] = C - 1  # what replace_negative_label does to the batch labels
        zz = z.clone().requires_grad_(True)
        j = torch.nn.CrossEntropyLoss(weight=cw)(zz, y2)
        j.backward()
        out[f"gb.{name}.logits"], out[f"gb.{name}.target"] = z.numpy(), y2.numpy()
        out[f"gb.{name}.csv_labels"], out[f"gb.{name}.class_weights"] = labels.astype(np.int64), cw.numpy()
        out[f"gb.{name}.loss"], out[f"gb.{name}.dlogits"] = j.detach().numpy(), zz.grad.numpy()
        names.append(name)
    out["gb.names"] = np.array(names)
    # the worked example of SURVEY.md Appendix B.5
    with tempfile.NamedTemporaryFile("w", suffix=".csv", delete=False) as f:
        for l, n in ((-1, 3), (0, 4), (1, 5), (2, 6)):
            for i in range(n):
                f.write(f"a/{l}_{i}.JPEG,{l}\n")
        path = f.name
    ds = ref_dataset.ImagenetDataset(path, "/nonexistent")
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ds.replace_negative_label()
    out["cw.example"] = ds.calculate_class_weights().numpy()
    os.unlink(path)
    # ---- confidence (reference metrics.py:8-42), both call styles of train.py:156-163 -----------------------
    g = torch.Generator().manual_seed(1)
    names = []
    for name, B, C, style in (("eos_c30", 16, 30, "eos"), ("bg_c31", 24, 31, "bg"), ("eos_noneg", 8, 30, "eos"), ("eos_nokn", 8, 30, "eos")):
        scores = torch.softmax(torch.randn(B, C, generator=g) * 2, dim=1)
        if style == "eos":
            y = torch.randint(-1, C, (B,), generator=g)
            if name == "eos_noneg":
                y = y.clamp(min=0)
            if name == "eos_nokn":
                y = torch.full((B,), -1)
            args = dict(offset=1.0 / C, unknown_class=-1, last_valid_class=None)
        else:
            y = torch.randint(0, C, (B,), generator=g)
            args = dict(offset=0.0, unknown_class=C - 1, last_valid_class=-1)
        r = ref_metrics.confidence(scores, y, **args)
        out[f"conf.{name}.scores"], out[f"conf.{name}.target"] = scores.numpy(), y.numpy()
        out[f"conf.{name}.args"] = np.array([args["offset"], args["unknown_class"], -999 if args["last_valid_class"] is None else args["last_valid_class"]], dtype=np.float64)
        out[f"conf.{name}.result"] = np.array(r, dtype=np.float64)
        names.append(name)
    out["conf.names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "losses_reference.npz"), **out)
    print("wrote losses_reference.npz with", len(out), "arrays")

    # ---- Part 2: whole-model vectors from the oracle restatement ---------------------------------------------
    from oracle import resnet50_oracle as R, losses_oracle as L
    vec = {}
    for tag, B, HW, C, seed in (("b8_128_c30", 8, 128, 30, 11), ("b16_96_c116", 16, 96, 116, 5)):
        for dt, dn in ((torch.float32, "f32"), (torch.float64, "f64")):
            gen = torch.Generator().manual_seed(seed)
            sd = R.init_state(C, C, False, generator=gen)
            x = torch.rand(B, 3, HW, HW, generator=gen)
            y = torch.randint(-1, C, (B,), generator=gen)
            sd = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
            logits, feats, loss, grads = R.forward_backward(sd, x.to(dt), y, lambda lg, t, f: L.entropic_openset_loss(lg, t, 1.0))
            p = f"{tag}.{dn}."
            vec[p + "logits"], vec[p + "features"], vec[p + "loss"] = logits.numpy(), feats.numpy(), loss.numpy()
            for k in ("logits.weight", "resnet_base.fc.bias", "resnet_base.layer1.0.conv1.weight", "resnet_base.layer1.0.bn1.weight",
                      "resnet_base.conv1.weight", "resnet_base.bn1.bias"):
                vec[p + "grad." + k] = grads[k].numpy()
            vec[p + "grad_norms"] = np.array([float(grads[k].double().norm()) for k in R.param_keys(sd)])
            vec[p + "bn1.running_mean"] = sd["resnet_base.bn1.running_mean"].numpy()
            vec[p + "bn1.running_var"] = sd["resnet_base.bn1.running_var"].numpy()
        vec[tag + ".meta"] = np.array([B, HW, C, seed])
    np.savez_compressed(os.path.join(HERE, "model_oracle.npz"), **vec)
    print("wrote model_oracle.npz with", len(vec), "arrays")


if __name__ == "__main__":
    main()
