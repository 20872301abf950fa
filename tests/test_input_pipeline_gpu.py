"""Device-side last mile of the input pipeline (next-row f3): uint8 [B,H,W,3] -> ToTensor (/255) + per-image horizontal flip +
NHWC4 staging in one kernel. Oracle: the documented torchvision semantics restated with torch ops (ToTensor of a uint8 image =
permute to CHW, .float().div(255); hflip = reverse the width axis) — torchvision itself is not installed, so this row is
'parity unpinned' at the third-party boundary like the ResNet body; the arithmetic is a single IEEE division and must be exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _to_tensor_hflip(u8, flip):
    x = u8.permute(0, 3, 1, 2).float().div(255)                      # transforms.ToTensor() on uint8 HWC
    return torch.where(torch.as_tensor(flip, dtype=torch.bool).view(-1, 1, 1, 1), x.flip(3), x)   # RandomHorizontalFlip outcome


@pytest.mark.parametrize("B,H,W", [(5, 64, 64), (3, 75, 91), (1, 224, 224)])
def test_u8_staging_kernel_is_exact(cuda, B, H, W):
    import osi_testlib as T
    from openset_imagenet import _native as N
    g = torch.Generator().manual_seed(B * H + W)
    u8 = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, generator=g)
    u8[0, 0, :256 if W >= 256 else W, 0] = torch.arange(min(W, 256), dtype=torch.uint8)   # every byte value appears
    flip = (torch.rand(B, generator=g) < 0.5).to(torch.uint8)
    flip[0] = 1
    ref = _to_tensor_hflip(u8, flip).permute(0, 2, 3, 1)               # NHWC
    ug, fg = u8.to(cuda), flip.to(cuda)
    y = torch.full((B, H, W, 4), float("nan"), device=cuda)
    N.check(N.lib().osi_u8hwc3_to_nhwc4(N.ptr(ug), N.ptr(fg), N.ptr(y), B, H, W, T.S()))
    assert torch.equal(y[..., :3].cpu(), ref) and float(y[..., 3].abs().max()) == 0
    N.check(N.lib().osi_u8hwc3_to_nhwc4(N.ptr(ug), None, N.ptr(y), B, H, W, T.S()))
    assert torch.equal(y[..., :3].cpu(), _to_tensor_hflip(u8, torch.zeros(B)).permute(0, 2, 3, 1))


def test_model_accepts_uint8_batches(cuda):
    """model(uint8 NHWC, flip) == model(ToTensor + flip on the host, fp32 NCHW): same logits and same gradients, bit for bit."""
    from openset_imagenet import ResNet50, EntropicOpensetLoss
    torch.manual_seed(3)
    C = 8
    model = ResNet50(C, C, False).to(cuda)
    g = torch.Generator().manual_seed(5)
    u8 = torch.randint(0, 256, (6, 96, 96, 3), dtype=torch.uint8, generator=g)
    flip = torch.tensor([1, 0, 0, 1, 1, 0], dtype=torch.uint8)
    y = torch.tensor([0, -1, 7, 3, -1, 2], device=cuda)
    loss = EntropicOpensetLoss(C, 1.0)
    outs = []
    for inp, fl in ((_to_tensor_hflip(u8, flip).to(cuda), None), (u8.to(cuda), flip)):
        model.train()
        lg, ft = model(inp) if fl is None else model(inp, flip=fl)
        loss(lg, y).backward()
        outs.append((lg.detach().clone(), ft.detach().clone(), model.flat_gradients().clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        model(u8.permute(0, 3, 1, 2).contiguous().to(cuda))        # uint8 must be [B,H,W,3]
    with pytest.raises(ValueError):
        model(outs[0][0].new_zeros(2, 3, 64, 64), flip=[0, 1])        # flip flags need the uint8 route


@pytest.mark.parametrize("u8", [True, False])
def test_worker_on_jpeg_files(cuda, tmp_path, u8):
    """The reference's data contract end to end (CSV of relative JPEG paths + labels, Resize(256) / crop / flip on the host
    workers, train.py:259-311) through worker(): with the default uint8 hand-over ToTensor runs on the GPU, with data.uint8 off
    the samples are fp32 CHW like the reference's; both train one epoch, validate, and write the checkpoints."""
    import os
    from PIL import Image
    from openset_imagenet import util
    from openset_imagenet.train import worker
    rng = np.random.default_rng(0)
    img_dir = tmp_path / "imgs"
    img_dir.mkdir()
    rows = {"train": [], "val": [], "test": []}
    for split, n in (("train", 10), ("val", 6), ("test", 7)):
        for i in range(n):
            h, w = int(rng.integers(260, 340)), int(rng.integers(260, 400))      # odd sizes: Resize(256) really resizes
            Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).save(img_dir / f"{split}_{i}.jpg", quality=90)
            rows[split].append(f"{split}_{i}.jpg,{-1 if i % 4 == 3 else i % 3}")
    proto = tmp_path / "protocols"
    proto.mkdir()
    (proto / "p2_train.csv").write_text("\n".join(rows["train"]) + "\n")
    (proto / "p2_val.csv").write_text("\n".join(rows["val"]) + "\n")
    (proto / "p2_test.csv").write_text("\n".join(rows["test"]) + "\n")
    cfg = util.load_yaml(os.path.join(os.path.dirname(__file__), "..", "config", "train.yaml"))
    cfg.epochs, cfg.batch_size, cfg.workers, cfg.parallel, cfg.gpu, cfg.protocol = 1, 4, 0, True, 0, 2
    cfg.loss.type = cfg.name = "entropic"
    cfg.data.imagenet_path = str(img_dir)
    cfg.data.train_file, cfg.data.val_file = str(proto / "p{}_train.csv"), str(proto / "p{}_val.csv")
    cfg.data.uint8 = u8
    cfg.output_directory = str(tmp_path / "out")
    best = worker(cfg)
    assert np.isfinite(best)
    ck = torch.load(tmp_path / "out" / "entropic_curr.pth", weights_only=False)
    assert ck["epoch"] == 1 and ck["model_state_dict"]["logits.weight"].shape[0] == 3      # 3 known classes, -1 = negatives
    if u8:   # the reference's evaluate.py surface on the checkpoint just written: arrays of both splits + the OSCR curve
        from openset_imagenet.script import evaluate
        from openset_imagenet.util import calculate_oscr
        from oracle.oscr_oracle import calculate_oscr as oracle_oscr
        files = evaluate.main(["entropic", "2", "-g", "0", "--imagenet-directory", str(img_dir), "--protocol-directory", str(proto),
                               "--output-directory", str(tmp_path / "out"), "--batch-size", "4", "--workers", "0", "--oscr"])
        for split, n in (("val", 6), ("test", 7)):
            a = np.load(files[split])
            assert a["gt"].shape == (n,) and a["logits"].shape == (n, 3) and a["features"].shape == (n, 3) and a["scores"].shape == (n, 3)
            assert np.allclose(a["scores"].sum(1), 1, atol=1e-5)
            ccr, fpr = calculate_oscr(a["gt"], a["scores"])
            occr, ofpr = oracle_oscr(a["gt"], a["scores"])
            assert np.array_equal(ccr, occr, equal_nan=True) and np.array_equal(fpr, ofpr, equal_nan=True)
