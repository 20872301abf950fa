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


@pytest.mark.parametrize("u8,workers", [(True, 0), (False, 0), (True, 2)])
def test_worker_on_jpeg_files(cuda, tmp_path, u8, workers):
    """The reference's data contract end to end (CSV of relative JPEG paths + labels, Resize(256) / crop / flip on the host
    workers, train.py:259-311) through worker(): with the default uint8 hand-over ToTensor runs on the GPU, with data.uint8 off
    the samples are fp32 CHW like the reference's; both train one epoch, validate, and write the checkpoints. With worker processes
    the loader builds sub-batches (data.sub_batches, default 4: here single samples) that the prefetcher groups back into batches
    of 4 — the ragged tails (10 = 2 * 4 + 2 training, 6 = 4 + 2 validation samples) included."""
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
    cfg.epochs, cfg.batch_size, cfg.workers, cfg.parallel, cfg.gpu, cfg.protocol = 1, 4, workers, True, 0, 2
    cfg.loss.type = cfg.name = "entropic"
    cfg.data.imagenet_path = str(img_dir)
    cfg.data.train_file, cfg.data.val_file = str(proto / "p{}_train.csv"), str(proto / "p{}_val.csv")
    cfg.data.uint8 = u8
    cfg.output_directory = str(tmp_path / "out")
    best = worker(cfg)
    assert np.isfinite(best)
    ck = torch.load(tmp_path / "out" / "entropic_curr.pth", weights_only=False)
    assert ck["epoch"] == 1 and ck["model_state_dict"]["logits.weight"].shape[0] == 3      # 3 known classes, -1 = negatives
    if u8 and not workers:   # the reference's evaluate.py surface on the checkpoint just written: arrays of both splits + the OSCR curve
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


def test_device_crop_flip_equals_pil(cuda):
    """The device side of Compose([Resize(256), RandomCrop(224) | CenterCrop(224), RandomHorizontalFlip, ToTensor]) (reference
    train.py:259-268) after decode + resize: the canvas kernel against PIL's own crop() / transpose(FLIP_LEFT_RIGHT) followed by
    the ToTensor division on the host — bit for bit, through the ctypes ABI and through torch.ops.osi.stage_canvas."""
    from PIL import Image
    from openset_imagenet import pipeline as P, _native as N
    rng = np.random.default_rng(7)
    canvases, crops, flips, refs = [], [], [], []
    for i, (w, h) in enumerate(((341, 256), (256, 300), (256, 256), (420, 256), (256, 511))):
        img = Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8))
        x0, y0 = int(rng.integers(0, w - 224 + 1)), int(rng.integers(0, h - 224 + 1))
        if i == 0:
            x0, y0 = w - 224, h - 224                                    # the far corner
        flip = bool(i % 2)
        ref = img.crop((x0, y0, x0 + 224, y0 + 224))                     # RandomCrop
        if flip:
            ref = ref.transpose(Image.FLIP_LEFT_RIGHT)                   # RandomHorizontalFlip acts on the cropped image
        refs.append(torch.from_numpy(np.asarray(ref).copy()).float().div(255))      # ToTensor (kept HWC for the comparison)
        win, cx, cy = P.canvas_window(np.asarray(img), x0, y0)
        canvases.append(torch.from_numpy(np.ascontiguousarray(win))); crops.append([cx, cy]); flips.append(int(flip))
    canvas = torch.stack(canvases).to(cuda)
    crop = torch.tensor(crops, dtype=torch.int32, device=cuda)
    flip = torch.tensor(flips, dtype=torch.uint8, device=cuda)
    ref = torch.stack(refs)
    out = P.stage_canvas_batch(canvas, crop, flip)
    assert out.shape == (5, 224, 224, 4) and torch.equal(out[..., :3].cpu(), ref) and float(out[..., 3].abs().max()) == 0
    out2 = N.ops().stage_canvas(canvas, crop, flip, 224, 224)
    assert torch.equal(out2, out)
    # no crop table / no flips = top-left corner, unflipped; out-of-range corners are clamped on the device, never read outside
    plain = N.ops().stage_canvas(canvas, None, None, 224, 224)
    assert torch.equal(plain[..., :3].cpu(), canvas[:, :224, :224].cpu().float().div(255))
    wild = torch.tensor([[-5, 999]] * 5, dtype=torch.int32, device=cuda)
    wild_before = wild.cpu().tolist()
    clamped = N.ops().stage_canvas(canvas, wild, None, 224, 224)
    assert torch.equal(clamped[..., :3].cpu(), canvas[:, 32:, :224].cpu().float().div(255)) and wild.cpu().tolist() == wild_before     # the caller's table is read-only


def test_prefetcher_equals_synchronous_loop(cuda):
    """DevicePrefetcher (copy stream, one batch ahead, event-ordered ring of staged batches): the batches it yields and a whole
    train() epoch driven by it equal the synchronous path bit for bit — a race between the copy and the compute stream (a ring
    slot rewritten while the stem weight gradient still reads it) would show up as different parameters."""
    from openset_imagenet import ResNet50, EntropicOpensetLoss, AverageMeter, optim, tools, pipeline as P
    from openset_imagenet.train import train
    from openset_imagenet.util import NameSpace
    tools.set_device_gpu(0)
    g = torch.Generator().manual_seed(3)
    n, C = 22, 5
    canv = torch.randint(0, 256, (n, 256, 256, 3), dtype=torch.uint8, generator=g)
    crop = torch.randint(0, 33, (n, 2), generator=g).to(torch.int32)
    flip = (torch.rand(n, generator=g) < 0.5).to(torch.uint8)
    lab = torch.randint(-1, C, (n,), generator=g)
    ds = torch.utils.data.TensorDataset(canv, crop, flip, lab)
    mk = lambda: torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, num_workers=0, pin_memory=True)   # ragged last batch of 2
    sync = [(P.stage_canvas_batch(c.to(cuda), xy.to(cuda), f.to(cuda)), y.to(cuda)) for c, xy, f, y in mk()]
    pre = [(x.clone(), y.clone()) for x, y in P.DevicePrefetcher(mk())]
    assert len(pre) == len(sync) == 6 and len(P.DevicePrefetcher(mk())) == 6
    for (a, ya), (b, yb) in zip(pre, sync):
        assert a.shape == b.shape and torch.equal(a, b) and torch.equal(ya, yb)

    def epoch(loader):
        torch.manual_seed(11)
        model = tools.device(ResNet50(C, C, False))
        opt = optim.Adam(model.parameters(), lr=1e-3)
        tr = {"j": AverageMeter()}
        train(model, loader, opt, EntropicOpensetLoss(C, 1.0), tr, NameSpace({"parallel": True}))
        torch.cuda.synchronize()
        return model.flat_parameters().clone(), model._flat_buffers.clone(), tr["j"].avg
    a = epoch(P.DevicePrefetcher(mk()))
    b = epoch([(x, y) for x, y in sync])
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2] and np.isfinite(a[2])

    # sub-batches (worker()'s data.sub_batches): loaders of batch 2 (shuffled by the same generator seed; one with worker processes)
    # grouped in pairs yield the batches of the batch-4 loader, ragged tail (22 = 5 * 4 + 2) included; so do plain (images, labels) pairs
    for nw in (0, 2):
        shuf = lambda bs, nw=nw: torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=True, num_workers=nw if bs == 2 else 0, pin_memory=True,
                                                             generator=torch.Generator().manual_seed(5))
        whole = [(x.clone(), y.clone()) for x, y in P.DevicePrefetcher(shuf(4))]
        fine = P.DevicePrefetcher(shuf(2), group=2)
        parts = [(x.clone(), y.clone()) for x, y in fine]
        assert len(fine) == len(parts) == len(whole) == 6 and parts[-1][0].shape[0] == 2
        for (a, ya), (b, yb) in zip(parts, whole):
            assert a.shape == b.shape and torch.equal(a, b) and torch.equal(ya, yb)
    pairs = torch.utils.data.TensorDataset(torch.randn(n, 3, 8, 8, generator=g), lab)
    whole = [(x.clone(), y.clone()) for x, y in P.DevicePrefetcher(torch.utils.data.DataLoader(pairs, batch_size=6))]
    parts = [(x.clone(), y.clone()) for x, y in P.DevicePrefetcher(torch.utils.data.DataLoader(pairs, batch_size=2), group=3)]
    assert len(parts) == len(whole) == 4 and all(torch.equal(a, b) and torch.equal(ya, yb) for (a, ya), (b, yb) in zip(parts, whole))
    with pytest.raises(ValueError):
        P.DevicePrefetcher(mk(), group=0)


def test_metrics_module_matches_reference_vectors(cuda, golden_dir):
    """metrics.confidence(scores, target, offset, unknown_class, last_valid_class) — the reference's own signature
    (metrics.py:8-42) — against the vectors its implementation produced; predict_objectosphere against its definition."""
    import os
    from openset_imagenet import metrics
    G = np.load(os.path.join(golden_dir, "losses_reference.npz"))
    for name in G["conf.names"]:
        p = f"conf.{name}."
        off, unk, last = G[p + "args"]
        r = metrics.confidence(torch.from_numpy(G[p + "scores"]).to(cuda), torch.from_numpy(G[p + "target"]).to(cuda), float(off),
                               int(unk), None if last == -999 else int(last))
        ref = G[p + "result"]
        assert (r[1], r[3]) == (ref[1], ref[3]) and abs(r[0] - ref[0]) < 1e-6 and abs(r[2] - ref[2]) < 1e-6, p
    g = torch.Generator().manual_seed(1)
    z, f = torch.randn(9, 6, generator=g), torch.randn(9, 6, generator=g)
    out = metrics.predict_objectosphere(z.to(cuda), f.to(cuda), 0.8).cpu()
    s = torch.softmax(z, 1)
    ps, pc = s.max(1)
    pc = pc.clone(); pc[(f.norm(dim=1) * ps) < 0.8] = -1
    assert torch.equal(out[:, 0].long(), pc) and torch.allclose(out[:, 1], ps, atol=1e-6)


def test_predict_objectosphere_matches_reference_vectors(cuda, golden_dir):
    """metrics.predict_objectosphere against vectors produced by the reference's own function (metrics.py:45-62;
    tests/golden/make_golden_misc.py): identical predicted classes (incl. the -1 rejections), scores to fp32 rounding."""
    import os
    from openset_imagenet import metrics
    G = np.load(os.path.join(golden_dir, "misc_reference.npz"))
    for name in G["po.names"]:
        z, f = torch.from_numpy(G[f"po.{name}.logits"]).to(cuda), torch.from_numpy(G[f"po.{name}.features"]).to(cuda)
        out = metrics.predict_objectosphere(z, f, float(G[f"po.{name}.threshold"])).cpu().numpy()
        ref = G[f"po.{name}.result"]
        assert out.shape == ref.shape and np.array_equal(out[:, 0], ref[:, 0]), name
        assert np.allclose(out[:, 1], ref[:, 1], rtol=2e-6, atol=1e-7), name


def test_sharded_validation_batches_through_the_prefetcher(cuda):
    """Validation on every rank, data side (reference loader: train.py:306-311): the validation DataLoader built on
    train.ShardedEvalBatches (rank r of `world`, sub-batches of batch / 4 that the prefetcher groups back, worker processes) yields
    exactly the batches r, r + world, ... of the plain unshuffled loader of the same batch size — ragged last batch included."""
    from openset_imagenet import tools, pipeline as P
    from openset_imagenet.train import ShardedEvalBatches
    tools.set_device_gpu(0)
    g = torch.Generator().manual_seed(9)
    n, B = 54, 8                                  # 7 batches, the last of 6 samples (two sub-batches: 2 + 2 + 2 -> 3 pieces of 2)
    canv = torch.randint(0, 256, (n, 256, 256, 3), dtype=torch.uint8, generator=g)
    crop = torch.randint(0, 33, (n, 2), generator=g).to(torch.int32)
    flip = torch.zeros(n, dtype=torch.uint8)
    lab = torch.arange(n)
    ds = torch.utils.data.TensorDataset(canv, crop, flip, lab)
    whole = [(x.clone(), y.clone()) for x, y in P.DevicePrefetcher(torch.utils.data.DataLoader(ds, batch_size=B, pin_memory=True))]
    assert len(whole) == 7 and whole[-1][1].numel() == 6
    for world, sub, nw in ((2, 4, 0), (4, 4, 2), (4, 1, 0), (8, 2, 0)):
        seen = {}
        for rank in range(world):
            sampler = ShardedEvalBatches(n, B, rank, world, sub)
            kw = dict(num_workers=nw, pin_memory=True)
            if nw:
                kw.update(persistent_workers=False, prefetch_factor=2)
            loader = P.DevicePrefetcher(torch.utils.data.DataLoader(ds, batch_sampler=sampler, **kw), group=sub)
            mine = [(x.clone(), y.clone()) for x, y in loader]
            assert len(mine) == len(sampler.batches()) == len(loader)
            for k, (x, y) in zip(sampler.batches(), mine):
                seen[k] = (x, y)
        assert sorted(seen) == list(range(7)), (world, sub)
        for k, (x, y) in seen.items():
            assert torch.equal(y, whole[k][1]) and torch.equal(x, whole[k][0]), (world, sub, k)
