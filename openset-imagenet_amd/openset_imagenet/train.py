"""Training loop, loss / optimizer construction and checkpointing with the call contract of the reference
openset_imagenet/train.py, on top of the MI355X executor.

  set_seeds(seed)                                            reference train.py:23-33
  save_checkpoint(f_name, model, epoch, opt, best_score_, scheduler=None)     train.py:37-60   (same dict schema)
  load_checkpoint(model, checkpoint, opt=None, scheduler=None) -> (start_epoch, best_score)   train.py:63-101
  train(model, data_loader, optimizer, loss_fn, trackers, cfg)                train.py:104-139 (same step order)
  build_loss / build_model / build_optimizer                                  train.py:329-369 (cfg.loss.type, cfg.opt.*)

Per-step order is the reference's: model.train(), zero_grad, H2D, forward, loss, tracker update, backward, step.
The one deliberate difference: the reference calls `j.item()` before `backward()` on every step (train.py:136), a full
device sync per step; here the loss scalars stay on the device and `trackers["j"]` receives exactly the same sequence
of `update(value, batch_len)` calls once, at the end of the epoch (identical avg / sum / count / val).
"""
import pathlib
import random
from collections import OrderedDict

import numpy as np
import torch

from . import dp as _dp
from . import losses as _losses
from . import optim as _optim
from . import tools
from .model import ResNet50


def set_seeds(seed):
    """Seed torch / random / numpy (reference train.py:23-33)."""
    torch.manual_seed(seed)
    random.seed(seed)
    np.random.seed(seed)


def _unwrap(model):
    return model.module if isinstance(model, (_dp.DistributedDataParallel, torch.nn.parallel.DistributedDataParallel)) else model


def save_checkpoint(f_name, model, epoch, opt, best_score_, scheduler=None):
    """Write {"epoch": epoch+1, "model_state_dict", "opt_state_dict", "best_score"[, "scheduler"]} (reference train.py:37-60)."""
    data = {"epoch": epoch + 1,
            "model_state_dict": _unwrap(model).state_dict(),
            "opt_state_dict": opt.state_dict(),
            "best_score": best_score_}
    if scheduler is not None:
        data["scheduler"] = scheduler.state_dict()
    torch.save(data, f_name)


def load_checkpoint(model, checkpoint, opt=None, scheduler=None):
    """Load a checkpoint written by this package or by the reference; strips a DDP "module." prefix
    (reference train.py:63-101). Returns (start_epoch, best_score); raises Exception when the file is missing."""
    file_path = pathlib.Path(checkpoint)
    if not file_path.is_file():
        raise Exception(f"Checkpoint file '{checkpoint}' not found")
    data = torch.load(file_path, map_location=tools.get_device(), weights_only=False)
    state = data["model_state_dict"]
    if list(state.keys())[0][:6] == "module":
        state = OrderedDict((k[7:], v) for k, v in state.items())
    _unwrap(model).load_state_dict(state)
    if opt is not None:
        opt.load_state_dict(data["opt_state_dict"])
    if scheduler is not None:
        scheduler.load_state_dict(data["scheduler"])
    return data["epoch"], data["best_score"]


def build_model(cfg, n_classes):
    """ResNet50(fc_layer_dim=n_classes, out_features=n_classes, logit_bias=False) on the global device (train.py:350-353)."""
    return tools.device(ResNet50(fc_layer_dim=n_classes, out_features=n_classes, logit_bias=False))


def build_loss(cfg, n_classes, class_weights=None):
    """cfg.loss.type in {entropic, softmax, garbage} as in the reference (train.py:339-347); `objectosphere` is this build's
    addition (keys loss.xi, loss.alpha)."""
    kind = cfg.loss.type
    if kind == "entropic":
        return _losses.EntropicOpensetLoss(n_classes, cfg.loss.w)
    if kind == "softmax":
        return _losses.SoftmaxLoss(ignore_index=-1)
    if kind == "garbage":
        if class_weights is None:
            raise ValueError("garbage loss needs the class weights of the training set (dataset.calculate_class_weights)")
        return _losses.GarbageLoss(tools.device(class_weights))
    if kind == "objectosphere":
        return _losses.ObjectosphereLoss(n_classes, cfg.loss.w, getattr(cfg.loss, "xi", 10.0), getattr(cfg.loss, "alpha", 1e-4))
    raise ValueError(f"unknown loss type {kind!r}")


def build_optimizer(cfg, model):
    """Adam(lr) or SGD(lr, momentum=0.9) over the model's arena (train.py:356-359)."""
    if cfg.opt.type == "sgd":
        return _optim.SGD(_unwrap(model), lr=cfg.opt.lr, momentum=0.9)
    return _optim.Adam(_unwrap(model), lr=cfg.opt.lr)


def train(model, data_loader, optimizer, loss_fn, trackers, cfg):
    """One epoch of training (reference train.py:104-139)."""
    for metric in trackers.values():
        metric.reset()
    if not cfg.parallel:
        import tqdm
        data_loader = tqdm.tqdm(data_loader)
    wants_features = isinstance(loss_fn, _losses.ObjectosphereLoss)
    pending, counts = [], []
    from .pipeline import device_batch
    for batch in data_loader:
        model.train()  # batch-norm uses and collects batch statistics
        optimizer.zero_grad()
        images, labels = device_batch(batch)   # device(images), device(labels) of train.py:128-129; canvas batches are staged
        batch_len = labels.shape[0]
        logits, features = model(images)
        j = loss_fn(logits, labels, features) if wants_features else loss_fn(logits, labels)
        pending.append(j.detach())
        counts.append(batch_len)
        j.backward()
        optimizer.step()
    if pending:
        for value, n in zip(torch.stack(pending).cpu().tolist(), counts):
            trackers["j"].update(value, n)


class ShardedEvalBatches(torch.utils.data.Sampler):
    """`batch_sampler` of the validation loader under data parallel. The reference's validation loader is
    `DataLoader(val_dataset, batch_size=cfg.batch_size)` (train.py:306-311): unshuffled, batch k = samples [k B, (k + 1) B), a ragged
    last batch kept. This sampler yields exactly those batches — but only the ones with k % world == rank, each cut into `sub`
    consecutive sub-batches of B / sub samples (what DevicePrefetcher(group=sub) reassembles; sub = 1: the batch itself). Whole
    batches stay whole, so every per-batch loss is the value the single-process loop computes for that batch."""

    def __init__(self, n_samples, batch_size, rank=0, world=1, sub=1):
        if batch_size < 1 or world < 1 or not 0 <= rank < world or sub < 1 or batch_size % sub:
            raise ValueError("ShardedEvalBatches: batch_size >= 1, 0 <= rank < world, batch_size % sub == 0")
        self.n, self.B, self.rank, self.world, self.step = int(n_samples), int(batch_size), int(rank), int(world), int(batch_size) // int(sub)

    def batches(self):
        """global indices of this rank's batches"""
        return range(self.rank, -(-self.n // self.B), self.world)

    def __iter__(self):
        for k in self.batches():
            lo, hi = k * self.B, min(self.n, (k + 1) * self.B)
            for s0 in range(lo, hi, self.step):
                yield list(range(s0, min(hi, s0 + self.step)))

    def __len__(self):
        return sum(-(-(min(self.n, (k + 1) * self.B) - k * self.B) // self.step) for k in self.batches())


def _confidence_partial(logits, labels, min_unk_score, unknown_class, last_valid, row):
    """row (double[4], zero) += {sum known score[y], #known, sum negatives (1 + offset - max score[:last_valid]), #negatives} of ONE
    batch: softmax + metrics.confidence (reference train.py:177, metrics.py:8-42) as one kernel, nothing synchronised."""
    from . import _native as N
    N.ops().confidence_accumulate(logits.contiguous(), labels.contiguous(), float(min_unk_score), int(unknown_class), int(last_valid), row)


def _broadcast_buffers(model, group=None):
    """BatchNorm running statistics of rank 0 on every rank (torch DDP's broadcast_buffers): the ranks normalise with their own batch
    statistics while training, so their running statistics drift apart; validation must score ONE model — the one rank 0 checkpoints."""
    import torch.distributed as dist
    m = _unwrap(model)
    if hasattr(m, "_flat_buffers"):
        dist.broadcast(m._flat_buffers, src=0, group=group)
        dist.broadcast(m._nbt, src=0, group=group)
    else:
        for b in m.buffers():
            dist.broadcast(b, src=0, group=group)


def validate(model, data_loader, loss_fn, n_classes, trackers, cfg, shard=None):
    """Validation loop with the reference's contract (train.py:142-196): eval-mode forward under no_grad, loss per batch into
    trackers["j"], known / negative confidences into trackers["conf_kn"] / ["conf_unk"].

    The reference fills an [N_val, C] softmax matrix on the device and reduces it with metrics.confidence() in Python loops
    (`sum(known)`, metrics.py:27-28). Here softmax + confidence are one kernel per batch that leaves the batch's four sums in a
    double[4] on the device; nothing is synchronised until the end of the loop, where the per-batch values are folded on the host in
    batch order — the very sequence of `update()` calls and additions the reference performs.

    `shard = (rank, world)` or `(rank, world, process_group)` (new — a departure from the reference's "Validate only on first
    process", train.py:248, which leaves world - 1 GPUs idle for a quarter of every epoch's samples, protocol.py:245-250): the loader
    yields only the batches rank, rank + world, ... of the unsharded, unshuffled batch sequence (ShardedEvalBatches); rank 0's BatchNorm
    buffers are broadcast first, every rank evaluates its batches, the per-batch (index, loss, count, confidence sums) records are
    all-gathered and EVERY rank replays them in batch order. The trackers are then bit-identical on all ranks to what a single
    process computes on the whole loader: whole batches, the same kernels, the same order of additions. A rank that fails reports
    its error through the same collective, so that no rank is left waiting: every rank raises."""
    for metric in trackers.values():
        metric.reset()
    if cfg.loss.type == "garbage":
        min_unk_score, unknown_class, last_valid = 0.0, n_classes - 1, -1
    else:
        min_unk_score, unknown_class, last_valid = 1.0 / n_classes, -1, 0   # 0 encodes Python's None (all columns)
    wants_features = isinstance(loss_fn, _losses.ObjectosphereLoss)
    rank, world, group = 0, 1, None
    if shard is not None:
        rank, world = int(shard[0]), int(shard[1])
        group = shard[2] if len(shard) > 2 else None
    model.eval()
    losses, counts, rows, chunks, records = [], [], [], [], []
    error = None
    try:
        if world > 1:
            _broadcast_buffers(model, group)
        from .pipeline import device_batch
        with torch.no_grad():
            for batch in data_loader:
                images, labels = device_batch(batch)
                logits, features = model(images)
                j = loss_fn(logits, labels, features) if wants_features else loss_fn(logits, labels)
                losses.append(j)
                counts.append(labels.shape[0])
                if len(rows) % 256 == 0:
                    chunks.append(torch.zeros(256, 4, dtype=torch.float64, device=logits.device))
                row = chunks[-1][len(rows) % 256]
                _confidence_partial(logits, labels, min_unk_score, unknown_class, last_valid, row)
                rows.append(row)
        if losses:                         # the one synchronisation of the loop (an asynchronous device error surfaces here: still inside the try)
            conf = torch.cat(chunks)[:len(rows)].cpu().tolist()
            records = [(rank + k * world, v, n, c) for k, (v, n, c) in enumerate(zip(torch.stack(losses).cpu().tolist(), counts, conf))]
    except Exception as e:                 # under data parallel the other ranks wait in the gather below: tell them there
        if world == 1:
            raise
        error, records = e, []
    if world > 1:
        import torch.distributed as dist
        everyone = [None] * world
        dist.all_gather_object(everyone, {"records": records, "error": None if error is None else f"{type(error).__name__}: {error}"}, group=group)
        if error is not None:
            raise error
        bad = [(r, e["error"]) for r, e in enumerate(everyone) if e["error"] is not None]
        if bad:
            raise RuntimeError(f"validate(): rank {bad[0][0]} failed: {bad[0][1]}")
        records = sorted(rec for e in everyone for rec in e["records"])
        if [rec[0] for rec in records] != list(range(len(records))):
            raise RuntimeError("validate(): the ranks' batches do not tile the batch sequence 0 .. n-1 — the loader is not sharded by "
                               "whole batches (ShardedEvalBatches(rank, world))")
    if not records:
        return
    acc = [0.0, 0.0, 0.0, 0.0]
    for _, value, n, c in records:         # batch order: AverageMeter sees the reference's update sequence, the sums its additions
        trackers["j"].update(value, n)
        for i in range(4):
            acc[i] += c[i]
    kn_sum, kn_count, neg_sum, neg_count = acc
    if kn_count:
        trackers["conf_kn"].update(kn_sum / kn_count, int(kn_count))
    if neg_count:
        trackers["conf_unk"].update(neg_sum / neg_count, int(neg_count))


def get_arrays(model, loader):
    """Targets, logits, deep features and softmax scores of a whole dataset as numpy arrays (reference train.py:200-234);
    everything is gathered on the device and copied to the host once."""
    from .pipeline import device_batch
    model.eval()
    t, lg, ft, sc = [], [], [], []
    with torch.no_grad():
        for batch in loader:
            images, labels = device_batch(batch)
            logit, feature = model(images)
            t.append(labels)
            lg.append(logit)
            ft.append(feature)
            sc.append(_losses.softmax(logit))
    cat = lambda xs: torch.cat(xs).cpu().numpy()
    return cat(t).astype(np.float32), cat(lg), cat(ft), cat(sc)


def _image_loader(csv_file, imagenet_path, train, loss_type, uint8=True):
    """ImagenetDataset of the reference (dataset.py:10-54) with its transforms (train.py:259-268) — see pipeline.CanvasDataset:
    the host workers decode and Resize(256); crop, flip, ToTensor and the layout staging run on the GPU (`uint8`, default), or the
    reference's own fp32 CHW samples are produced on the host (`uint8=False`)."""
    from .pipeline import CanvasDataset
    return CanvasDataset(csv_file, imagenet_path, train, loss_type, uint8)


def dist_env():
    """(rank, world, local_rank) of a process started by `python -m torch.distributed.run` (or by the CLI's own launcher,
    script/train.py) — (0, 1, None) otherwise. The reference never initialises a process group (its `dist:` block is unused,
    config/train.yaml:35-39); this is what the block was for: one process per GPU, batch_size per GPU (train.yaml:18)."""
    import os
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        return int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
    return 0, 1, None


_last_worker_state = {}   # introspection for tests: the objects of the most recent worker() call in this process


def worker(cfg):
    """Creates datasets, model, loss, optimizer, runs the epoch loop with checkpoints — the reference's worker()
    (train.py:237-482) reduced to what drives the hot path. Same cfg keys (config/train.yaml), same checkpoint files
    `{name}_curr.pth` / `{name}_best.pth`, same best-score rule (conf_kn + conf_unk) and early stopping. Logging goes to the
    `logging` module and a CSV of per-epoch scalars (loguru / TensorBoard are not dependencies of this build).

    Data parallel (new; the reference only left vestiges, train.py:10,49-50,79-87,248): under `torch.distributed.run` (or the
    CLI launcher driven by the `dist:` block) every rank runs this function on its own GPU (LOCAL_RANK), the training set is
    sharded with a DistributedSampler, the model is wrapped in dp.DistributedDataParallel (bucketed RCCL gradient all-reduce
    overlapped with backward), `batch_size` is per GPU (train.yaml:18), only rank 0 logs and writes checkpoints ("Log only on first
    process", train.py:248) — and, departing from "Validate only on first process" (same line), EVERY rank validates its share of the
    batches (validate(shard=...), `dist.shard_validation`, default on; off = the reference's rule).

    `cfg.data.synthetic` (new key, default absent) = number of synthetic training samples to use instead of the CSV files."""
    import logging
    import torch.distributed as dist
    from . import _native
    _native.lib()
    if _native.DIAGNOSTIC_LIB:     # OSI_HIP_LIB + OSI_DEV=1: ablated / instrumented builds, some wrong by design — never a training run
        raise RuntimeError(f"worker(): refusing to train on the diagnostic native library {_native.DIAGNOSTIC_LIB}; unset OSI_HIP_LIB")
    set_seeds(cfg.seed)
    rank, world, local_rank = dist_env()
    distributed = world > 1
    out_dir = pathlib.Path(cfg.output_directory)
    out_dir.mkdir(parents=True, exist_ok=True)
    handlers = [logging.StreamHandler()]
    if rank == 0:
        handlers.append(logging.FileHandler(out_dir / cfg.log_name, mode="w"))
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING,
                        format="%(asctime)s %(name)s %(levelname)s: %(message)s", handlers=handlers, force=True)
    log = logging.getLogger("openset_imagenet")
    dcfg = getattr(cfg, "dist", None)
    backend = getattr(dcfg, "backend", None) or "nccl"
    if distributed:
        # one process per GPU: the launcher started us before any GPU call; LOCAL_RANK picks the device (several ranks may share
        # a device only over gloo, which is how the one-GPU test box rehearses this path)
        n_dev = max(1, torch.cuda.device_count())
        index = local_rank % n_dev if backend != "nccl" else local_rank
        dev = tools.set_device_gpu(index=index)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    elif cfg.gpu is not None:
        tools.set_device_gpu(index=cfg.gpu)
    else:
        raise RuntimeError("No GPU device selected: the MI355X build has no CPU training path (pass -g [index])")
    try:
        return _worker_body(cfg, log, out_dir, rank, world, distributed)
    finally:
        if distributed and dist.is_initialized():
            dist.destroy_process_group()


def _worker_body(cfg, log, out_dir, rank, world, distributed):
    import time
    import torch.distributed as dist
    from .dataset import LabelTable, SyntheticImagenet
    from .pipeline import DevicePrefetcher
    n_syn = getattr(cfg.data, "synthetic", None)
    if n_syn:
        n_known = {1: 116, 2: 30, 3: 151}[int(cfg.protocol)]
        g = torch.Generator().manual_seed(cfg.seed)
        def labels(n):
            y = torch.randint(0, n_known, (n,), generator=g)
            y[torch.rand(n, generator=g) < 0.4] = -1
            return y
        tables = []
        for n, is_train in ((int(n_syn), True), (max(int(n_syn) // 4, cfg.batch_size), False)):
            t = LabelTable(labels(n).numpy())
            if cfg.loss.type == "garbage":
                t.replace_negative_label()
            elif cfg.loss.type == "softmax" and is_train:
                t.remove_negative_label()
            tables.append(t)
        train_table, val_table = tables
        train_ds, val_ds = SyntheticImagenet(train_table.labels, seed=1), SyntheticImagenet(val_table.labels, seed=2)
    else:
        train_file = pathlib.Path(cfg.data.train_file.format(cfg.protocol))
        val_file = pathlib.Path(cfg.data.val_file.format(cfg.protocol))
        if not (train_file.exists() and val_file.exists()):
            raise FileNotFoundError("train/validation file does not exist")
        u8 = bool(getattr(cfg.data, "uint8", True))   # new key: False = fp32 CHW samples exactly as the reference's loader yields
        train_ds = _image_loader(train_file, cfg.data.imagenet_path, True, cfg.loss.type, u8)
        val_ds = _image_loader(val_file, cfg.data.imagenet_path, False, cfg.loss.type, u8)
        train_table = train_ds.table
    # loaders (train.py:299-311). batch_size is per GPU; under data parallel every rank draws its own shard of each epoch
    sampler = torch.utils.data.distributed.DistributedSampler(train_ds, num_replicas=world, rank=rank, shuffle=True, seed=cfg.seed) \
        if distributed else None
    prefetch = bool(getattr(cfg.data, "prefetch", True))   # new key: copy-stream prefetch + device-side crop / flip / ToTensor
    # new key data.sub_batches: with the prefetcher and worker processes, the workers build batch_size / sub_batches samples at a time
    # and the prefetcher stages sub_batches consecutive ones into one step's batch — the same batches (a sampler's index stream is only
    # cut finer), a 1/sub_batches as long wait for the first batch of every epoch (profiles/r05_input_pipeline.json)
    sub = int(getattr(cfg.data, "sub_batches", 4)) if (prefetch and cfg.workers > 0) else 1
    if sub < 1 or cfg.batch_size % sub:
        sub = 1
    lkw = dict(batch_size=cfg.batch_size // sub, num_workers=cfg.workers, pin_memory=True)
    if cfg.workers > 0:
        from .pipeline import worker_init
        lkw.update(persistent_workers=True, prefetch_factor=4 * sub,   # the same number of samples ahead per worker
                   worker_init_fn=worker_init)                         # one intra-op thread per decode worker
    train_loader = torch.utils.data.DataLoader(train_ds, shuffle=sampler is None, sampler=sampler, **lkw)
    # validation (new key dist.shard_validation, default on): under data parallel every rank scores the batches rank, rank + world, ...
    # of the unshuffled validation sequence (whole batches: validate() replays them in order, bit-identical trackers); off = the
    # reference's comment "Validate only on first process" (train.py:248)
    shard_val = distributed and bool(getattr(getattr(cfg, "dist", None), "shard_validation", True))
    if shard_val:
        vkw = {k: v for k, v in lkw.items() if k != "batch_size"}
        val_loader = torch.utils.data.DataLoader(val_ds, batch_sampler=ShardedEvalBatches(len(val_ds), cfg.batch_size, rank, world, sub), **vkw)
    else:
        val_loader = torch.utils.data.DataLoader(val_ds, **lkw)
    if prefetch:
        train_loader, val_loader = DevicePrefetcher(train_loader, group=sub), DevicePrefetcher(val_loader, group=sub)

    # number of classes / loss (train.py:329-347): entropic has no output for the unknown label
    n_classes = train_table.label_count - 1 if cfg.loss.type == "entropic" else train_table.label_count
    class_weights = train_table.calculate_class_weights() if cfg.loss.type == "garbage" else None
    loss_fn = build_loss(cfg, n_classes, class_weights)
    model = build_model(cfg, n_classes)
    opt = build_optimizer(cfg, model)
    scheduler = torch.optim.lr_scheduler.StepLR(opt, step_size=cfg.opt.decay, gamma=cfg.opt.gamma) if cfg.opt.decay > 0 else None

    best_score, start_epoch = 0.0, 0
    if cfg.checkpoint is not None:
        if cfg.train_mode == "finetune":     # weights only; keeps the checkpoint's epoch, resets the best score (train.py:374-380)
            start_epoch, _ = load_checkpoint(model, cfg.checkpoint)
        else:
            start_epoch, best_score = load_checkpoint(model, cfg.checkpoint, opt, scheduler)
        log.info(f"Loaded {cfg.checkpoint} at epoch {start_epoch}")
    net = _dp.DistributedDataParallel(model) if distributed else model   # broadcasts rank 0's parameters and BN buffers
    _last_worker_state.clear()
    _last_worker_state.update(model=model, optimizer=opt, rank=rank, world=world, checkpoints_written=0, sharded_validation=shard_val)
    t_metrics = {"j": _losses.AverageMeter()}
    v_metrics = {"j": _losses.AverageMeter(), "conf_kn": _losses.AverageMeter(), "conf_unk": _losses.AverageMeter()}
    _last_worker_state.update(v_metrics=v_metrics, val_loader=val_loader, loss_fn=loss_fn, n_classes=n_classes)
    early = _losses.EarlyStopping(patience=cfg.patience) if cfg.patience > 0 else None
    scalars = None
    if rank == 0:
        scalars = open(out_dir / f"scalars-{cfg.log_name}.csv", "w")
        scalars.write("epoch,train/loss,val/loss,val/conf_kn,val/conf_unk\n")
    log.info(f"Training: protocol {cfg.protocol}, loss {cfg.loss.type}, {n_classes} classes, {len(train_ds)} / {len(val_ds)} samples, "
             f"{world} GPU(s) x batch {cfg.batch_size}")
    for epoch in range(start_epoch, cfg.epochs):
        t0 = time.time()
        if sampler is not None:
            sampler.set_epoch(epoch)
        train(net, train_loader, opt, loss_fn, t_metrics, cfg)
        t1 = time.time()
        stop, failure = False, None
        curr_score = None
        if shard_val:                                     # every rank scores its share of the batches; all ranks end with the same trackers
            validate(model, val_loader, loss_fn, n_classes, v_metrics, cfg, shard=(rank, world))
            curr_score = v_metrics["conf_kn"].avg + v_metrics["conf_unk"].avg
        elif rank == 0:                                   # validate on the first process only (reference train.py:248)
            try:
                validate(model, val_loader, loss_fn, n_classes, v_metrics, cfg)
                curr_score = v_metrics["conf_kn"].avg + v_metrics["conf_unk"].avg
            except Exception as e:                        # the other ranks wait in the broadcast below: tell them before re-raising.
                failure = e                               # (KeyboardInterrupt / SystemExit propagate at once: the launcher ends the ranks)
                log.exception("rank 0 failed in validate()")   # the original traceback, before any collective can mask it
        # learning-rate schedule: stepped on every rank, after validation and BEFORE the checkpoints are written (reference
        # train.py:435-437 then :463-471), so that `_curr.pth` / `_best.pth` carry the scheduler state and learning rate of the epoch
        # they resume into
        if scheduler is not None and failure is None:
            scheduler.step()
        if rank == 0 and failure is None:
            try:
                scalars.write(f"{epoch},{t_metrics['j'].avg},{v_metrics['j'].avg},{v_metrics['conf_kn'].avg},{v_metrics['conf_unk'].avg}\n")
                scalars.flush()
                log.info(f"ep:{epoch} train:{t_metrics} val:{v_metrics} t:{t1 - t0:.1f}s v:{time.time() - t1:.1f}s")
                save_checkpoint(out_dir / (cfg.name + "_curr.pth"), model, epoch, opt, curr_score, scheduler)
                _last_worker_state["checkpoints_written"] += 1
                if curr_score > best_score:
                    best_score = curr_score
                    save_checkpoint(out_dir / (cfg.name + "_best.pth"), model, epoch, opt, best_score, scheduler)
                    _last_worker_state["checkpoints_written"] += 1
                if early is not None:
                    early(metrics=curr_score, loss=False)
                    stop = early.early_stop
            except Exception as e:
                failure = e
                log.exception("rank 0 failed while logging / writing checkpoints")
        if distributed:                                   # every rank follows rank 0's early-stopping decision — or its failure
            flag = [stop, best_score, None if failure is None else repr(failure)]
            dist.broadcast_object_list(flag, src=0)
            stop, best_score, remote_failure = flag
            if rank != 0 and remote_failure is not None:
                raise RuntimeError(f"rank 0 failed in its validation / checkpoint section: {remote_failure}")
        if failure is not None:
            raise failure
        if stop:
            log.info("early stop")
            break
    if scalars is not None:
        scalars.close()
    return best_score
