"""Independent witness for the ResNet-50 body of the oracle (dev container only; needs `transformers`, not the GPU box).

The reference builds its body with `torchvision.models.resnet50` (reference openset_imagenet/model.py:17), which is neither
vendored under /root/reference nor installed here. oracle/resnet50_oracle.py restates that topology from the published
definition. This script checks the restatement against a SECOND third-party implementation of the same published network:
`transformers.models.resnet.ResNetModel` configured as ResNet-50 v1.5 (bottleneck layers, depths 3/4/6/3, stride on the 3x3:
`downsample_in_bottleneck=False`, no stride in the first stage). transformers shares no code with the oracle; the only
thing taken from the oracle is the WEIGHT VALUES (seeded `init_state` + `randomize_bn`), copied into the transformers
module through an explicit key map, so that the test can regenerate them without transformers.

Outputs (tests/golden/resnet_witness.npz): for each case, in float64, train mode and eval mode:
  pooled [B, 2048]  = ResNetModel(...).pooler_output           (the body: stem, 16 bottlenecks, global average pool)
  features, logits  = the reference head (model.py:19-26,37-39) applied to `pooled` with torch.nn.functional.linear
  bn running_mean / running_var of three BatchNorms after the train-mode forward (momentum / unbiased-variance rule)
  gradients (round 3): autograd is enabled on the transformers module in train mode and a fixed scalar — the entropic open-set
  loss (reference losses.py:16-29, closed form in oracle/losses_oracle.py, itself pinned to the reference's vectors) of the
  witness logits with seeded labels — is back-propagated through it (the reference side: j.backward(), train.py:138). Committed in
  float64: the L2 norm of all 162 parameter gradients, the full gradient of conv1, of conv1 / bn1 of block 0 of every stage (the
  two largest as strided samples, see GRAD_FULL / GRAD_SAMPLED), of the last BatchNorm, of fc and of the logits layer. The oracle's
  backward graph is thereby pinned by an implementation that shares no code with it.
tests/test_oracle.py::test_oracle_body_matches_transformers_witness compares oracle.forward / forward_backward against these arrays.

Run: python tests/golden/make_golden_witness.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

from oracle import resnet50_oracle as R  # noqa: E402

CASES = (("b4_64_c30", 4, 64, 30, 101), ("b2_96_c116", 2, 96, 116, 102))
WATCHED_BN = ("resnet_base.bn1", "resnet_base.layer2.0.downsample.1", "resnet_base.layer4.2.bn3")
GRAD_FULL = ("resnet_base.conv1.weight", "resnet_base.bn1.weight", "resnet_base.bn1.bias",
             "resnet_base.layer1.0.conv1.weight", "resnet_base.layer1.0.bn1.weight", "resnet_base.layer1.0.bn1.bias",
             "resnet_base.layer1.0.downsample.0.weight", "resnet_base.layer2.0.bn1.weight",
             "resnet_base.layer2.3.bn2.bias", "resnet_base.layer3.0.bn1.weight", "resnet_base.layer3.5.bn3.weight",
             "resnet_base.layer4.0.bn1.bias", "resnet_base.layer4.2.bn3.weight", "resnet_base.layer4.2.bn3.bias",
             "resnet_base.fc.bias", "logits.weight")
GRAD_SAMPLED = {"resnet_base.fc.weight": 5, "resnet_base.layer2.0.conv1.weight": 3, "resnet_base.layer3.0.conv1.weight": 7, "resnet_base.layer3.2.conv2.weight": 53, "resnet_base.layer4.0.conv1.weight": 31,
                "resnet_base.layer4.1.conv2.weight": 211, "resnet_base.layer4.0.downsample.0.weight": 97}   # every n-th element, flattened OIHW


def hf_key(k):
    """reference / torchvision state_dict key -> transformers ResNetModel key (body only)."""
    k = k[len("resnet_base."):]
    tail = {"weight": "weight", "bias": "bias", "running_mean": "running_mean", "running_var": "running_var",
            "num_batches_tracked": "num_batches_tracked"}
    parts = k.split(".")
    if parts[0] == "conv1":
        return "embedder.embedder.convolution.weight"
    if parts[0] == "bn1":
        return "embedder.embedder.normalization." + tail[parts[1]]
    stage, block, leaf = int(parts[0][5:]) - 1, int(parts[1]), parts[2]
    pre = f"encoder.stages.{stage}.layers.{block}."
    if leaf == "downsample":
        return pre + "shortcut." + ("convolution.weight" if parts[3] == "0" else "normalization." + tail[parts[4]])
    idx = int(leaf[-1]) - 1
    return pre + f"layer.{idx}." + ("convolution.weight" if leaf.startswith("conv") else "normalization." + tail[parts[3]])


def main():
    from transformers import ResNetConfig, ResNetModel
    out = {"names": np.array([c[0] for c in CASES])}
    for tag, B, HW, C, seed in CASES:
        gen = torch.Generator().manual_seed(seed)
        sd = R.init_state(C, C, False, generator=gen)
        R.randomize_bn(sd, gen)
        x = torch.rand(B, 3, HW, HW, generator=gen)
        cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3],
                           layer_type="bottleneck", hidden_act="relu", downsample_in_first_stage=False,
                           downsample_in_bottleneck=False)
        hf = ResNetModel(cfg).double()
        body = {hf_key(k): v.double() if v.is_floating_point() else v.clone() for k, v in sd.items() if k.startswith("resnet_base.")
                and not k.startswith("resnet_base.fc.")}
        missing, unexpected = hf.load_state_dict(body, strict=True)
        assert not missing and not unexpected
        fcw, fcb, lw = sd["resnet_base.fc.weight"].double(), sd["resnet_base.fc.bias"].double(), sd["logits.weight"].double()
        for mode in ("train", "eval"):
            hf.load_state_dict(body, strict=True)      # fresh running statistics for each mode
            hf.train(mode == "train")
            with torch.no_grad():
                pooled = hf(x.double()).pooler_output.flatten(1)
            feats = F.linear(pooled, fcw, fcb)
            logits = F.linear(feats, lw)
            p = f"{tag}.{mode}."
            out[p + "pooled"], out[p + "features"], out[p + "logits"] = pooled.clone().numpy(), feats.clone().numpy(), logits.clone().numpy()
            if mode == "train":
                hsd = hf.state_dict()
                for bn in WATCHED_BN:
                    out[p + bn + ".running_mean"] = hsd[hf_key(bn + ".running_mean")].clone().numpy()   # state_dict tensors alias the buffers
                    out[p + bn + ".running_var"] = hsd[hf_key(bn + ".running_var")].clone().numpy()
                    assert int(hsd[hf_key(bn + ".num_batches_tracked")]) == 1
        # ---- gradients: the transformers module under autograd, train mode, the entropic loss of its logits -------------------
        from oracle import losses_oracle as L
        hf.load_state_dict(body, strict=True)
        hf.train(True)
        for q in hf.parameters():
            q.requires_grad_(True)
            q.grad = None
        head = {"resnet_base.fc.weight": fcw.clone().requires_grad_(True), "resnet_base.fc.bias": fcb.clone().requires_grad_(True),
                "logits.weight": lw.clone().requires_grad_(True)}
        y = torch.randint(-1, C, (B,), generator=torch.Generator().manual_seed(seed + 1000))
        pooled = hf(x.double()).pooler_output.flatten(1)
        logits = F.linear(F.linear(pooled, head["resnet_base.fc.weight"], head["resnet_base.fc.bias"]), head["logits.weight"])
        loss = L.entropic_openset_loss(logits, y, 1.0)
        loss.backward()
        hfp = dict(hf.named_parameters())
        grads = {}
        for k in R.param_keys(sd):
            grads[k] = head[k].grad if k in head else hfp[hf_key(k)].grad
            assert grads[k] is not None and grads[k].shape == sd[k].shape, k
        keys = R.param_keys(sd)
        assert len(keys) == 162
        out[tag + ".grad.labels"] = y.numpy()
        out[tag + ".grad.loss"] = np.array(float(loss))
        out[tag + ".grad.norms"] = np.array([float(grads[k].norm()) for k in keys])
        for k in GRAD_FULL:
            out[f"{tag}.grad.full.{k}"] = grads[k].detach().numpy().copy()
        for k, n in GRAD_SAMPLED.items():
            out[f"{tag}.grad.every{n}.{k}"] = grads[k].detach().flatten()[::n].numpy().copy()
        out[tag + ".meta"] = np.array([B, HW, C, seed])
        print(tag, "pooled", tuple(out[f"{tag}.train.pooled"].shape), "max |logit|", float(np.abs(out[f"{tag}.train.logits"]).max()))
    np.savez_compressed(os.path.join(HERE, "resnet_witness.npz"), **out)
    print("wrote resnet_witness.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
