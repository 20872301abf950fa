"""CPU: host-side mirror of the reference interface — module tree / state_dict / checkpoint dict / optimizer state / config /
meters — and the loud refusal to compute without a GPU."""
import os

import numpy as np
import pytest
import torch

import openset_imagenet as oi
from openset_imagenet import optim, tools, util
from openset_imagenet.train import load_checkpoint, save_checkpoint, build_loss
from oracle import resnet50_oracle as R


@pytest.fixture(scope="module")
def model():
    torch.manual_seed(0)
    return oi.ResNet50(fc_layer_dim=12, out_features=12, logit_bias=False)


def test_state_dict_is_the_reference_layout(model):
    sd = model.state_dict()
    assert list(sd) == R.state_keys()
    ref = R.init_state(12, 12)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(ref[k].shape), k
        assert v.dtype == ref[k].dtype, k
    assert model.logits.in_features == 12 and model.logits.out_features == 12       # read at reference train.py:210-211
    assert sum(p.numel() for p in model.parameters()) == 23454912 + 53120 + 2048 * 12 + 12 + 144
    # conv weights: logical OIHW, physical KRSC (what the kernels read)
    w = model.resnet_base.layer1._modules["0"].conv2.weight
    assert w.shape == (64, 64, 3, 3) and w.stride() == (576, 1, 192, 64)
    assert float(model.resnet_base.bn1.weight.min()) == 1.0 and float(model.resnet_base.bn1.running_var.max()) == 1.0
    std = float(w.std())
    assert abs(std - (2.0 / (64 * 9)) ** 0.5) < 0.1 * std                        # kaiming-normal, fan_out


def test_load_reference_format_state_dict_and_arena_sharing(model):
    ref = R.init_state(12, 12, generator=torch.Generator().manual_seed(3))
    model.load_state_dict(ref)
    for k, v in model.state_dict().items():
        assert torch.equal(v, ref[k]), k
    # parameters are views of one arena: writing the arena changes the parameter
    flat = model.flat_parameters()
    p = model.resnet_base.conv1.weight
    assert p.data_ptr() == flat.data_ptr()
    flat[0] = 42.0
    assert float(p[0, 0, 0, 0]) == 42.0
    # DDP-prefixed checkpoints load too (reference train.py:79-87)
    pref = {"module." + k: v for k, v in ref.items()}
    torch.save({"epoch": 3, "model_state_dict": pref, "opt_state_dict": {}, "best_score": 0.5}, "/tmp/_osi_ddp.pth")
    m2 = oi.ResNet50(12, 12, False)
    assert load_checkpoint(m2, "/tmp/_osi_ddp.pth") == (3, 0.5)
    assert torch.equal(m2.state_dict()["logits.weight"], ref["logits.weight"])
    with pytest.raises(Exception):
        load_checkpoint(m2, "/tmp/does_not_exist.pth")


def test_checkpoint_round_trip_and_optimizer_state(model, tmp_path):
    opt = optim.Adam(model.parameters(), lr=2e-3)
    f = tmp_path / "m_curr.pth"
    save_checkpoint(f, model, 6, opt, 1.25, scheduler=torch.optim.lr_scheduler.StepLR(opt, 5, 0.1))
    data = torch.load(f, weights_only=False)
    assert set(data) == {"epoch", "model_state_dict", "opt_state_dict", "best_score", "scheduler"} and data["epoch"] == 7
    assert list(data["model_state_dict"]) == R.state_keys()
    assert data["opt_state_dict"]["param_groups"][0]["lr"] == 2e-3
    # the parameter list of the reference spelling is accepted, anything else is refused
    optim.SGD(params=model.parameters(), lr=0.1)
    with pytest.raises(ValueError):
        optim.Adam([torch.nn.Parameter(torch.zeros(4))], lr=1e-3)
    # a stock torch optimizer state loads into the fused one (same state_dict schema)
    twin = [torch.nn.Parameter(p.detach().clone()) for p in model.parameters()]
    t = torch.optim.Adam(twin, lr=1e-3)
    for q in twin:
        q.grad = torch.ones_like(q)
    t.step()
    opt.load_state_dict(t.state_dict())
    assert opt._steps == 1
    assert torch.allclose(opt.state[model._plist[5]]["exp_avg"], t.state[twin[5]]["exp_avg"])


def test_no_cpu_fallback(model):
    with pytest.raises(RuntimeError, match="no CPU path"):
        model(torch.rand(1, 3, 64, 64))
    with pytest.raises(RuntimeError):
        oi.EntropicOpensetLoss(4)(torch.randn(2, 4, requires_grad=True), torch.tensor([0, -1]))
    with pytest.raises(ValueError):
        model(torch.rand(1, 1, 64, 64))


def test_config_meters_and_loss_factory(tmp_path):
    y = tmp_path / "train.yaml"
    y.write_text("name: x\nparallel: false\nloss:\n  type: entropic\n  w: 2.0\nopt:\n  type: adam\n  lr: 1.e-3\nbatch_size: 64\n")
    cfg = util.load_yaml(y)
    assert cfg.loss.type == "entropic" and cfg.opt.lr == 1e-3 and cfg.batch_size == 64
    assert util.load_yaml(y).dump() == cfg.dump() and "loss" in cfg.dict()
    m = oi.AverageMeter()
    m.update(2.0, 4); m.update(4.0, 12)
    assert m.avg == 3.5 and m.count == 16 and m.val == 4.0 and repr(m) == "3.500"
    es = oi.EarlyStopping(patience=2)
    for v in (1.0, 1.1, 1.2):
        es(v, loss=True)
    assert es.early_stop
    assert isinstance(build_loss(cfg, 30), oi.EntropicOpensetLoss) and build_loss(cfg, 30).unk_weight == 2.0
    cfg.loss.type = "garbage"
    with pytest.raises(ValueError):
        build_loss(cfg, 30)
    assert tools.device(torch.zeros(1)).device.type == "cpu" and tools.get_device().type == "cpu"


def test_cli_arguments_and_label_table():
    from openset_imagenet.script.train import get_args
    from openset_imagenet.dataset import LabelTable, SyntheticImagenet
    a = get_args(["cfg.yaml", "2", "-g", "--nice", "0"])
    assert a.gpu == 0 and a.protocol == 2 and str(a.output_directory) == "."       # bare -g selects GPU 0 (reference bug fixed)
    assert get_args(["cfg.yaml", "1", "--nice", "0"]).gpu is None
    t = LabelTable([-1, -1, -1, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2])
    assert t.has_negatives() and t.label_count == 4
    t.replace_negative_label()
    assert np.allclose(t.calculate_class_weights().numpy(), [1.125, 0.9, 0.75, 1.5])
    t2 = LabelTable([-1, 0, 1, 1]); t2.remove_negative_label()
    assert t2.label_count == 2 and list(t2.labels) == [0, 1, 1]
    ds = SyntheticImagenet([3, -1], image_size=32)
    x, y = ds[1]
    assert x.shape == (3, 32, 32) and x.dtype == torch.float32 and 0 <= float(x.min()) and float(x.max()) < 1 and int(y) == -1
    assert torch.equal(ds[1][0], x)
