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
    tools.set_device_cpu()       # (a GPU test earlier in the same process may have selected cuda:0)
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


def test_package_surface_matches_the_reference_init():
    """The reference's openset_imagenet/__init__.py:1-7 exports ImagenetDataset, ResNet50 and the sub-modules util, train,
    metrics, losses (OpenSetProtocol is out of scope and documented as such)."""
    for name in ("ResNet50", "ImagenetDataset", "util", "train", "metrics", "losses"):
        assert hasattr(oi, name), name
    assert callable(oi.train.load_checkpoint) and callable(oi.train.worker) and callable(oi.metrics.confidence)
    assert callable(oi.metrics.predict_objectosphere) and callable(oi.losses.EntropicOpensetLoss)
    with pytest.raises(RuntimeError):                       # the metric kernels need the GPU too
        oi.metrics.confidence(torch.rand(2, 3), torch.tensor([0, -1]))


def test_torch_op_library_registers_without_a_gpu():
    """libosi_torch.so (TORCH_LIBRARY(osi, ...)) loads on a CPU-only host and registers every hot-path op; CPU tensors find no
    kernel (the ops are registered for the HIP device key only) — no compute happens here."""
    from openset_imagenet import _native as N
    ops = N.ops()
    for name in ("resnet50_forward", "resnet50_backward", "loss_fwd_bwd", "adam_step", "sgd_step", "stage_canvas", "softmax",
                 "confidence_accumulate"):
        assert hasattr(ops, name), name
    with pytest.raises(NotImplementedError):
        ops.softmax(torch.zeros(2, 3))
    assert N.lib().osi_abi_version() >= 5


def test_tuning_knobs_are_explicit():
    """Development switches travel through osi_set_tuning / osi_resnet50_set_option, not through getenv inside launch functions."""
    import ctypes
    from openset_imagenet import _native as N
    lib = N.lib()
    v = ctypes.c_int()
    assert lib.osi_get_tuning(b"wgrad_blocks", ctypes.byref(v)) == 0 and v.value == 2048
    assert lib.osi_set_tuning(b"wgrad_blocks", 1024) == 0 and lib.osi_get_tuning(b"wgrad_blocks", ctypes.byref(v)) == 0 and v.value == 1024
    assert lib.osi_set_tuning(b"wgrad_blocks", 2048) == 0
    assert lib.osi_set_tuning(b"wgrad_nst", 3) == -1 and lib.osi_set_tuning(b"no_such_knob", 1) == -1
    h = ctypes.c_void_p()
    assert lib.osi_resnet50_create(ctypes.byref(h), 2, 64, 64, 5, 5, 0) == 0
    try:
        assert lib.osi_resnet50_set_option(h, b"overlap", 0) == 0 and lib.osi_resnet50_set_option(h, b"fwd_fork", 0) == 0
        assert lib.osi_resnet50_set_option(h, b"side_priority_normal", 1) == 0
        assert lib.osi_resnet50_set_option(h, b"bogus", 1) == -1
        assert lib.osi_resnet50_bind_input_nhwc4(h, None) == -1 and lib.osi_resnet50_bind_input_nhwc4(h, 24) == -1   # NULL / unaligned
    finally:
        lib.osi_resnet50_destroy(h)
    src = "".join(open(os.path.join(os.path.dirname(__file__), "..", "openset-imagenet_amd", "csrc", f)).read()
                  for f in ("conv_igemm.hip", "stem_direct.hip", "bn.hip", "resnet50_exec.hip", "pool_layout.hip", "loss.hip", "optim.hip", "linear.hip"))
    assert "getenv" not in src


def test_host_side_of_the_input_pipeline(tmp_path):
    """Resize(256) / CenterCrop(224) size rules and the canvas window (host logic of pipeline.CanvasDataset) against PIL."""
    from PIL import Image
    from openset_imagenet import pipeline as P
    assert P.resize_size(500, 375) == (341, 256) and P.resize_size(375, 500) == (256, 341) and P.resize_size(256, 256) == (256, 256)
    assert P.resize_size(1000, 333) == (int(256 * 1000 / 333), 256)                       # torchvision: int(), not round()
    assert P.center_crop_corner(341, 256) == (int(round(117 / 2.0)), 16) == (58, 16)      # Python rounding (banker's) like torchvision
    assert P.center_crop_corner(343, 256)[0] == int(round(119 / 2.0)) == 60
    rng = np.random.default_rng(1)
    arr = rng.integers(0, 256, size=(256, 341, 3), dtype=np.uint8)
    for x0, y0 in ((0, 0), (117, 32), (60, 7), (100, 31)):
        win, cx, cy = P.canvas_window(arr, x0, y0)
        assert win.shape == (256, 256, 3) and 0 <= cx <= 32 and 0 <= cy <= 32
        assert np.array_equal(win[cy:cy + 224, cx:cx + 224], arr[y0:y0 + 224, x0:x0 + 224])
    # the dataset object end to end on two JPEG files (CPU part only: decode, resize, window, labels)
    rows = []
    for i, (w, h) in enumerate(((400, 300), (280, 390))):
        Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).save(tmp_path / f"{i}.jpg", quality=95)
        rows.append(f"{i}.jpg,{i - 1}")
    (tmp_path / "p.csv").write_text("\n".join(rows) + "\n")
    ds = oi.CanvasDataset(tmp_path / "p.csv", tmp_path, train=False, loss_type="entropic")
    assert len(ds) == 2 and ds.table.label_count == 2 and ds.table.has_negatives()
    canvas, crop, flip, label = ds[0]
    assert canvas.shape == (256, 256, 3) and canvas.dtype == torch.uint8 and crop.dtype == torch.int32 and int(flip) == 0 and int(label) == -1
    img = Image.open(tmp_path / "0.jpg").convert("RGB").resize((341, 256), Image.BILINEAR)
    x0, y0 = P.center_crop_corner(341, 256)
    ref = np.asarray(img.crop((x0, y0, x0 + 224, y0 + 224)))
    cx, cy = (int(v) for v in crop)
    assert np.array_equal(canvas.numpy()[cy:cy + 224, cx:cx + 224], ref)
    x, y = oi.CanvasDataset(tmp_path / "p.csv", tmp_path, train=False, loss_type="entropic", uint8=False)[0]   # the reference's own sample
    assert x.shape == (3, 224, 224) and torch.equal(x, torch.from_numpy(ref.copy()).permute(2, 0, 1).float().div(255))
    garbage = oi.CanvasDataset(tmp_path / "p.csv", tmp_path, train=True, loss_type="garbage")
    assert int(garbage[0][3]) == 1 and garbage.table.label_count == 2     # -1 relabelled to the last index (dataset.py:60-68)


def test_dist_environment_and_launcher(monkeypatch, tmp_path):
    from openset_imagenet.train import dist_env
    from openset_imagenet.script import train as cli
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert dist_env() == (0, 1, None)
    monkeypatch.setenv("RANK", "3"); monkeypatch.setenv("WORLD_SIZE", "8"); monkeypatch.setenv("LOCAL_RANK", "3")
    assert dist_env() == (3, 8, 3)
    # the reference's `dist:` block is back in the configuration, default off
    cfg = util.load_yaml(os.path.join(os.path.dirname(__file__), "..", "config", "train.yaml"))
    assert cfg.dist.distributed is False and cfg.dist.gpus == 2 and str(cfg.dist.port) == "8889"
    # distributed: on + no RANK in the environment -> the CLI becomes the launcher (one child per GPU, before any GPU call)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    cfg.dist.distributed = True
    y = tmp_path / "d.yaml"
    y.write_text(cfg.dump())
    started = []

    class FakeProc:
        def __init__(self, cmd, env):
            started.append((cmd, {k: env[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}))

        def poll(self):
            return 0
    monkeypatch.setattr(cli.subprocess, "Popen", lambda cmd, env: FakeProc(cmd, env))
    assert cli.main([str(y), "2", "-g", "--nice", "0"]) == 0
    assert [e["RANK"] for _, e in started] == ["0", "1"] and all(e["WORLD_SIZE"] == "2" and e["MASTER_ADDR"] == "127.0.0.1" and
                                                               e["MASTER_PORT"] == "8889" for _, e in started)
    assert started[0][0][1:3] == ["-m", "openset_imagenet.script.train"] and started[0][0][3:] == [str(y), "2", "-g", "--nice", "0"]


def test_launcher_terminates_the_surviving_ranks_when_one_dies():
    """A rank that dies must not leave its siblings parked in a collective until the process-group timeout: the launcher polls,
    terminates the exact children it started on the first non-zero exit and reports every exit code."""
    import subprocess, sys, time
    from openset_imagenet.script import train as cli
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, "-c", "import time; time.sleep(120)"]),
             subprocess.Popen([sys.executable, "-c", "import sys, time; time.sleep(0.3); sys.exit(3)"])]
    with pytest.raises(SystemExit) as e:
        cli._wait_ranks(procs, poll_seconds=0.05, grace_seconds=5.0)
    assert time.time() - t0 < 30 and all(p.poll() is not None for p in procs)
    assert "3" in str(e.value) and "-15" in str(e.value)          # the dead rank's code and the survivor's SIGTERM
    ok = [subprocess.Popen([sys.executable, "-c", "pass"]) for _ in range(2)]
    assert cli._wait_ranks(ok, poll_seconds=0.05) == 0


def test_imagenet_dataset_has_the_reference_constructor_and_members(tmp_path):
    """`ImagenetDataset(csv_file, imagenet_path, transform=None)` as reference-style callers use it (reference dataset.py:13-86):
    (image, label) samples with the transform applied to the PIL image, label_count counting the -1 class (dataset.py:26), the two
    label rewrites and the class weights of SURVEY Appendix B.5 (18 / (count * 4) -> [1.125, 0.9, 0.75, 1.5])."""
    from PIL import Image
    rng = np.random.default_rng(1)
    labels = [-1] * 3 + [0] * 4 + [1] * 5 + [2] * 6
    rows = []
    for i, lab in enumerate(labels):
        Image.fromarray(rng.integers(0, 256, size=(20, 24, 3), dtype=np.uint8)).save(tmp_path / f"{i}.png")
        rows.append(f"{i}.png,{lab}")
    (tmp_path / "d.csv").write_text("\n".join(rows) + "\n")
    to_tensor = lambda im: torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).float().div(255)
    ds = oi.ImagenetDataset(tmp_path / "d.csv", tmp_path, transform=to_tensor)
    assert len(ds) == 18 and ds.label_count == 4 and list(ds.unique_classes) == [-1, 0, 1, 2] and ds.has_negatives()
    x, y = ds[torch.tensor(0)]
    assert x.shape == (3, 20, 24) and y.dtype == torch.int64 and int(y) == -1
    assert oi.ImagenetDataset(tmp_path / "d.csv", tmp_path)[3][0].size == (24, 20)        # no transform: the PIL image itself
    ds.replace_negative_label()
    assert list(ds.unique_classes) == [0, 1, 2, 3] and int(ds[0][1]) == 3 and ds.label_count == 4 and not ds.has_negatives()
    assert np.allclose(ds.calculate_class_weights().numpy(), [1.125, 0.9, 0.75, 1.5])
    ds2 = oi.ImagenetDataset(tmp_path / "d.csv", tmp_path)
    ds2.remove_negative_label()
    assert len(ds2) == 15 and ds2.label_count == 3 and list(ds2.unique_classes) == [0, 1, 2] and int(ds2[0][1]) == 0


def test_bench_record_helpers(tmp_path):
    """bench.py's host-side helpers for the N > 1 record: RCCL's channel count parsed from its own INIT log (both line forms, the last
    communicator wins, suffixed log files are found), the CPU share of a lease (x the GPUs THIS RUN uses, never the visible count),
    the CPU model string."""
    import importlib.util
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    spec = importlib.util.spec_from_file_location("osi_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    log = tmp_path / "rccl.log"
    (tmp_path / "rccl.log.host.123").write_text("x:1:2 [0] NCCL INFO Channel 00/128 : 0\nx:1:2 [0] NCCL INFO Channel 01/128 : 0\n")
    info = bench.rccl_channels(str(log))
    assert info["coll_channels"] == 128 and "Channel" in info["source"]
    log.write_text("h:9:9 [0] NCCL INFO 16 coll channels, 16 collnet channels, 0 nvls channels, 32 p2p channels\n"
                   "h:9:9 [0] NCCL INFO 32 coll channels, 32 collnet channels, 0 nvls channels, 32 p2p channels, 2 p2p channels per peer\n")
    assert bench.rccl_channels(str(log))["coll_channels"] == 32
    assert bench.rccl_channels(None)["coll_channels"] is None and bench.rccl_channels(str(tmp_path / "missing"))["coll_channels"] is None
    one, eight = bench.usable_cpus(1, 16), bench.usable_cpus(8, 16)
    assert one["lease_share"] == 16 and eight["lease_share"] == 128 and one["threads"] <= 16 and 1 <= one["threads"] <= eight["threads"]
    assert bench.usable_cpus(1, 2)["threads"] <= 2
    assert isinstance(bench.cpu_model(), str) and bench.cpu_model()
